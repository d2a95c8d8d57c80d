// The diagnostic instances of the whole-layer kernel K8 (rqs_resnet_kernel.hpp; DBG = true; round 6): 8 bins, ReLU
// blocks, no context, the plain final-layer loop, with one more store per tile of the run's LAST layer -- the final
// Linear's accumulators, i.e. the conditioner's output as the spline evaluation reads it (the width / height rows
// already divided by sqrt(hidden_features), coupling.py:554-556).  tests/test_gpu_logits.py holds them to the error an
// fp32 library GEMM has against float64.  Reached through nfa_rqs_flow_resnet_logits_f32 only.
#include "rqs_resnet_kernel.hpp"

namespace nfa {

ResnetKernelFn resnet_debug_kernel(bool inverse, int init_ks) {
    if (init_ks == 4)
        return inverse ? rqs_resnet_kernel<true, 1, 4, 0, 8, false, kActRelu, true>
                       : rqs_resnet_kernel<false, 1, 4, 0, 8, false, kActRelu, true>;
    return inverse ? rqs_resnet_kernel<true, 1, 2, 0, 8, false, kActRelu, true>
                   : rqs_resnet_kernel<false, 1, 2, 0, 8, false, kActRelu, true>;
}

}  // namespace nfa
