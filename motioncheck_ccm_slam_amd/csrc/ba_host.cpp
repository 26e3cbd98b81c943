// placeholder until the BA path lands (next commit)
#include "ccm_internal.h"
struct BaState {};
void ba_state_free(BaState* s) { delete s; }
extern "C" int ccm_ba_solve(ccm_ctx* c, ccm_ba_problem*, const ccm_ba_options*, ccm_ba_result*) { return ccm_fail(c, CCM_E_STATE, "BA not built"); }
extern "C" int ccm_pose_from_mat4f(const float*, double*) { return CCM_E_STATE; }
extern "C" int ccm_pose_to_mat4f(const double*, float*) { return CCM_E_STATE; }
