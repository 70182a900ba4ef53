/* dev helper (round 6): set_structure + analyze through the C ABI from a plain C process (no Python, no reference): is the
 * analysis as fast here as tools/cold_path.py measures it from Python?  Structure file: tools/dump_structure.py.
 * gcc -O2 -I include -o analyze_c analyze_c.c -L slam_plus_plus_amd -lslampp_hip -Wl,-rpath,$PWD/slam_plus_plus_amd */
#include "slampp_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include <unistd.h>

static double now_ms(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char **argv)
{
	if(argc < 2)
		return 1;
	FILE *f = fopen(argv[1], "rb");
	if(!f)
		return 2;
	int64_t n;
	if(fread(&n, 8, 1, f) != 1)
		return 3;
	int64_t *cumsum = malloc((n + 1) * 8), *ptr = malloc((n + 1) * 8);
	if(fread(cumsum, 8, n + 1, f) != (size_t)(n + 1) || fread(ptr, 8, n + 1, f) != (size_t)(n + 1))
		return 3;
	int32_t *brow = malloc(ptr[n] * 4);
	if(fread(brow, 4, ptr[n], f) != (size_t)ptr[n])
		return 3;
	fclose(f);
	const int reps = (argc > 2)? atoi(argv[2]) : 4;
	for(int r = 0; r < reps + 1; ++ r) {
		slampp_hip_solver *p = 0;
		if(slampp_hip_create(&p, 0) != SLAMPP_HIP_OK)
			return 4;
		usleep(30000);
		const double t0 = now_ms();
		int rc = slampp_hip_set_structure(p, n, cumsum, ptr, brow);
		if(rc == SLAMPP_HIP_OK)
			rc = slampp_hip_analyze(p, SLAMPP_HIP_MODE_SPARSE, 0);
		const double t1 = now_ms();
		printf("%s: set_structure + analyze %.2f ms (status %d)%s\n", argv[1], t1 - t0, rc, r? "" : "  <- the process's first handle");
		slampp_hip_destroy(p);
	}
	return 0;
}
