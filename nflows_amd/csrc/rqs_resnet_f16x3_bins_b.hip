// Instances of the whole-layer kernel K8x (rqs_resnet_f16x3_kernel.hpp; design notes in rqs_resnet_f16x3.hip) for the bin
// counts other than 8 (13, 14, 15, 16, 20, 24, 32): a translation unit of their own.
#include "rqs_resnet_f16x3_kernel.hpp"

namespace nfa {
namespace k8x {

#define NFA_K8X_PICK(KB_)                                                                                                  \
    (init_ks == 4 ? (inverse ? rqs_resnet_f16x3_kernel<true, 4, false, KB_> : rqs_resnet_f16x3_kernel<false, 4, false, KB_>) \
                  : (inverse ? rqs_resnet_f16x3_kernel<true, 2, false, KB_> : rqs_resnet_f16x3_kernel<false, 2, false, KB_>))

KernelFn bins_kernel_b(int K, bool inverse, int init_ks) {
    switch (K) {
        case 13: return NFA_K8X_PICK(13);
        case 14: return NFA_K8X_PICK(14);
        case 15: return NFA_K8X_PICK(15);
        case 16: return NFA_K8X_PICK(16);
        case 20: return NFA_K8X_PICK(20);
        case 24: return NFA_K8X_PICK(24);
        case 32: return NFA_K8X_PICK(32);
    }
    return nullptr;
}

}  // namespace k8x
}  // namespace nfa
