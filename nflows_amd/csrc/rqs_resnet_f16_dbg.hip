// The diagnostic instances of the whole-layer kernel K8h (rqs_resnet_f16_kernel.hpp; DBG = true): the bench's kernel
// family (8 bins, ReLU blocks, no context) with one more store per evaluation of the run's LAST layer -- the bin the
// evaluation's single walk chose (FusedSteps::kbin).  K8h builds its knots as fp32 running sums and keeps no bin
// index; these instances make "which bin did it pick" observable so that tests/test_gpu_bin_index.py can compare it with
// torchutils.searchsorted on the reference's knots (utils/torchutils.py:134-136, rational_quadratic.py:115-118) and hold
// the elements where the two differ to the output tolerances.  Reached through nfa_rqs_flow_resnet_f16x2_bins_f32 only.
#include "rqs_resnet_f16_kernel.hpp"

namespace nfa {
namespace k8h {

KernelFn debug_kernel(bool inverse, int init_ks, int waves) {
    if (init_ks != 2) return nullptr;   // (d_i <= 32)
    if (waves == 8)
        return inverse ? rqs_resnet_f16_kernel<true, 2, 8, 8, false, kRing, kActRelu, true>
                       : rqs_resnet_f16_kernel<false, 2, 8, 8, false, kRing, kActRelu, true>;
    return inverse ? rqs_resnet_f16_kernel<true, 2, 4, 8, false, kRing, kActRelu, true>
                   : rqs_resnet_f16_kernel<false, 2, 4, 8, false, kRing, kActRelu, true>;
}

}  // namespace k8h
}  // namespace nfa
