// Instances of the whole-layer kernel K8h (rqs_resnet_f16_kernel.hpp; design notes in rqs_resnet_f16.hip) for 2 .. 7 and 9 bins:
// a translation unit of their own so that the library's ~200 instances of that kernel compile side by side.
#include "rqs_resnet_f16_kernel.hpp"

namespace nfa {
namespace k8h {

#define NFA_K8H_PICK(KB_, ACT_)                                                                                      \
    (waves == 8 ? (init_ks == 4 ? (inverse ? rqs_resnet_f16_kernel<true, 4, 8, KB_, false, kRing, ACT_>                 \
                                           : rqs_resnet_f16_kernel<false, 4, 8, KB_, false, kRing, ACT_>)                \
                                : (inverse ? rqs_resnet_f16_kernel<true, 2, 8, KB_, false, kRing, ACT_>                 \
                                           : rqs_resnet_f16_kernel<false, 2, 8, KB_, false, kRing, ACT_>))               \
                : (init_ks == 4 ? (inverse ? rqs_resnet_f16_kernel<true, 4, 4, KB_, false, kRing, ACT_>                 \
                                           : rqs_resnet_f16_kernel<false, 4, 4, KB_, false, kRing, ACT_>)                \
                                : (inverse ? rqs_resnet_f16_kernel<true, 2, 4, KB_, false, kRing, ACT_>                 \
                                           : rqs_resnet_f16_kernel<false, 2, 4, KB_, false, kRing, ACT_>)))

KernelFn bins_kernel_a(int K, bool inverse, int init_ks, int waves) {
    switch (K) {
        case 2: return NFA_K8H_PICK(2, kActRelu);
        case 3: return NFA_K8H_PICK(3, kActRelu);
        case 4: return NFA_K8H_PICK(4, kActRelu);
        case 5: return NFA_K8H_PICK(5, kActRelu);
        case 6: return NFA_K8H_PICK(6, kActRelu);
        case 7: return NFA_K8H_PICK(7, kActRelu);
        case 9: return NFA_K8H_PICK(9, kActRelu);
    }
    return nullptr;
}

}  // namespace k8h
}  // namespace nfa
