// Instances of the whole-layer kernel K8h (rqs_resnet_f16_kernel.hpp; design notes in rqs_resnet_f16.hip) for 11 .. 16 bins:
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

KernelFn bins_kernel_b(int K, bool inverse, int init_ks, int waves) {
    switch (K) {
        case 11: return NFA_K8H_PICK(11, kActRelu);
        case 12: return NFA_K8H_PICK(12, kActRelu);
        case 13: return NFA_K8H_PICK(13, kActRelu);
        case 14: return NFA_K8H_PICK(14, kActRelu);
        case 15: return NFA_K8H_PICK(15, kActRelu);
        case 16: return NFA_K8H_PICK(16, kActRelu);
    }
    return nullptr;
}

}  // namespace k8h
}  // namespace nfa
