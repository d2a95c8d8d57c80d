// Instances of the whole-layer kernel K8x (rqs_resnet_f16x3_kernel.hpp; design notes in rqs_resnet_f16x3.hip) for the bin
// counts other than 8 (2, 3, 4, 5, 6, 7, 9, 10, 11, 12): a translation unit of their own.
#include "rqs_resnet_f16x3_kernel.hpp"

namespace nfa {
namespace k8x {

#define NFA_K8X_PICK(KB_)                                                                                                  \
    (init_ks == 4 ? (inverse ? rqs_resnet_f16x3_kernel<true, 4, false, KB_> : rqs_resnet_f16x3_kernel<false, 4, false, KB_>) \
                  : (inverse ? rqs_resnet_f16x3_kernel<true, 2, false, KB_> : rqs_resnet_f16x3_kernel<false, 2, false, KB_>))

KernelFn bins_kernel_a(int K, bool inverse, int init_ks) {
    switch (K) {
        case 2: return NFA_K8X_PICK(2);
        case 3: return NFA_K8X_PICK(3);
        case 4: return NFA_K8X_PICK(4);
        case 5: return NFA_K8X_PICK(5);
        case 6: return NFA_K8X_PICK(6);
        case 7: return NFA_K8X_PICK(7);
        case 9: return NFA_K8X_PICK(9);
        case 10: return NFA_K8X_PICK(10);
        case 11: return NFA_K8X_PICK(11);
        case 12: return NFA_K8X_PICK(12);
    }
    return nullptr;
}

}  // namespace k8x
}  // namespace nfa
