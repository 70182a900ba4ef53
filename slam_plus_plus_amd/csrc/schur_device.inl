// schur_device.inl -- device helpers shared by schur.hip and schur_tiles.hip
#pragma once

// small SPD inverse through its Cholesky factor; returns false on a non-positive pivot
template <int D>
__device__ __forceinline__ bool spd_inverse(const double *a /* column-major, upper triangle read */, double *inv)
{
	double L[D][D], X[D][D];
	bool ok = true;
	#pragma unroll
	for(int j = 0; j < D; ++ j) {
		double s = a[j + j * D];
		#pragma unroll
		for(int k = 0; k < D; ++ k)
			if(k < j) s -= L[j][k] * L[j][k];
		if(!(s > 0)) { ok = false; s = 1; }
		const double d = sqrt(s);
		L[j][j] = d;
		#pragma unroll
		for(int i = 0; i < D; ++ i) {
			if(i > j) {
				double t = a[j + i * D]; // element (j, i) of the upper triangle = (i, j)
				#pragma unroll
				for(int k = 0; k < D; ++ k)
					if(k < j) t -= L[i][k] * L[j][k];
				L[i][j] = t / d;
			}
		}
	}
	// X = L^-1 (lower)
	#pragma unroll
	for(int c = 0; c < D; ++ c) {
		#pragma unroll
		for(int r = 0; r < D; ++ r) {
			if(r < c) X[r][c] = 0;
			else if(r == c) X[r][c] = 1.0 / L[r][r];
			else {
				double t = 0;
				#pragma unroll
				for(int k = 0; k < D; ++ k)
					if(k >= c && k < r) t += L[r][k] * X[k][c];
				X[r][c] = -t / L[r][r];
			}
		}
	}
	// inv = X^T X
	#pragma unroll
	for(int c = 0; c < D; ++ c)
		#pragma unroll
		for(int r = 0; r < D; ++ r) {
			double t = 0;
			#pragma unroll
			for(int k = 0; k < D; ++ k)
				if(k >= r && k >= c) t += X[k][r] * X[k][c];
			inv[r + c * D] = t;
		}
	return ok;
}
