// schur.hip -- placeholder, replaced below
#include "solver.h"
namespace slampp {
struct CSchurState {};
void schur_destroy(CSchurState *p) { delete p; }
CSchurState *schur_analyze(slampp_hip_solver &) { throw std::domain_error("Schur path not built yet"); }
void schur_enqueue(slampp_hip_solver &, const double *, double *) { throw std::domain_error("Schur path not built yet"); }
size_t schur_device_bytes(const CSchurState *) { return 0; }
void schur_fill_stats(const CSchurState *, slampp_hip_stats &) {}
}
