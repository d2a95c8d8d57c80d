// Instances of the whole-layer kernel K8h (rqs_resnet_f16_kernel.hpp; design notes in rqs_resnet_f16.hip) for 20, 24 and 32 bins, and the block activations other than ReLU (8 / 10 bins):
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

KernelFn bins_kernel_c(int K, bool inverse, int init_ks, int waves) {
    switch (K) {
        case 20: return NFA_K8H_PICK(20, kActRelu);
        case 24: return NFA_K8H_PICK(24, kActRelu);
        case 32: return NFA_K8H_PICK(32, kActRelu);
    }
    return nullptr;
}

KernelFn activation_kernel(int activation, int K, bool inverse, int init_ks, int waves) {
    if (K != 8 && K != 10) return nullptr;
    switch (activation) {
        case NFA_ACTIVATION_LEAKY_RELU: return K == 8 ? NFA_K8H_PICK(8, kActLeakyRelu) : NFA_K8H_PICK(10, kActLeakyRelu);
        case NFA_ACTIVATION_ELU: return K == 8 ? NFA_K8H_PICK(8, kActElu) : NFA_K8H_PICK(10, kActElu);
        case NFA_ACTIVATION_TANH: return K == 8 ? NFA_K8H_PICK(8, kActTanh) : NFA_K8H_PICK(10, kActTanh);
    }
    return nullptr;
}

}  // namespace k8h
}  // namespace nfa
