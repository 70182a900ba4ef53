/*
 * oracle/ba_scene.h -- TEST INFRASTRUCTURE (oracle), not product code.
 *
 * A synthetic BA scene in the reference's own types (CVertexCam / CVertexXYZ / CEdgeP2C3D, include/slam/BA_Types.h):
 * cameras on a ring looking at a cloud of points around the origin, every point observed by n_obs_per_point cameras;
 * measurements are the reference's own projection of the true scene (CBAJacobians::Project_P2C) plus pixel noise, the
 * initial state is the truth perturbed.  Shared by oracle/dropin_driver.cpp (the reference's LM solver on
 * CLinearSolver_HIP) and oracle/ref_harness.cpp (`lambda_dump ba_lm`: the Lambda that LM solver hands to its linear solver).
 */
#pragma once
#include <random>
#include <vector>
#include <math.h>

template <class CSystemType>
static void Build_BA_Scene(CSystemType &system, size_t n_cams, size_t n_points, size_t n_obs_per_point, unsigned n_seed)
{
	std::mt19937_64 rng(n_seed);
	std::normal_distribution<double> nd(0, 1);
	std::vector<Eigen::Matrix<double, 6, 1> > cams(n_cams);
	Eigen::Matrix<double, 5, 1> intrinsics;
	intrinsics << 500, 500, 320, 240, 0;
	for(size_t c = 0; c < n_cams; ++ c) {
		const double a = 2 * M_PI * double(c) / double(n_cams);
		Eigen::Vector3d C(8 * cos(a), 8 * sin(a), 1.5 * sin(3 * a)); // camera centre
		Eigen::Vector3d z = (-C).normalized(), x = Eigen::Vector3d(0, 0, 1).cross(z).normalized(), y = z.cross(x);
		Eigen::Matrix3d R;
		R.row(0) = x; R.row(1) = y; R.row(2) = z; // world -> camera
		cams[c].head<3>() = -R * C;
		cams[c].tail<3>() = C3DJacobians::v_RotMatrix_to_AxisAngle(R);
	}
	std::vector<Eigen::Vector3d> points(n_points);
	for(size_t p = 0; p < n_points; ++ p)
		points[p] = Eigen::Vector3d(1.5 * nd(rng), 1.5 * nd(rng), 1.0 * nd(rng));
	for(size_t c = 0; c < n_cams; ++ c) { // cameras first: ids 0 .. n_cams - 1
		Eigen::Matrix<double, 11, 1> v;
		v.head<6>() = cams[c];
		v.tail<5>() = intrinsics;
		for(int d = 0; d < 3 && c > 0; ++ d) { // (camera 0 stays at the truth)
			v(d) += 0.02 * nd(rng);
			v(3 + d) += 0.005 * nd(rng);
		}
		system.template r_Get_Vertex<CVertexCam>(c, v);
	}
	for(size_t p = 0; p < n_points; ++ p) {
		Eigen::Vector3d v = points[p];
		for(int d = 0; d < 3; ++ d)
			v(d) += 0.03 * nd(rng);
		system.template r_Get_Vertex<CVertexXYZ>(n_cams + p, v);
	}
	const Eigen::Matrix2d information = Eigen::Matrix2d::Identity();
	for(size_t p = 0; p < n_points; ++ p) {
		const size_t c0 = rng() % n_cams;
		for(size_t k = 0; k < n_obs_per_point; ++ k) {
			const size_t c = (c0 + k * (n_cams / n_obs_per_point)) % n_cams; // distinct while n_obs_per_point <= n_cams
			Eigen::Vector2d uv;
			CBAJacobians::Project_P2C(cams[c], intrinsics, points[p], uv);
			uv(0) += 0.3 * nd(rng);
			uv(1) += 0.3 * nd(rng);
			system.r_Add_Edge(CEdgeP2C3D(n_cams + p, c, uv, information, system)); // (xyz id, camera id, ...)
		}
	}
}
