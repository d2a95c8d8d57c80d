// Instances of the whole-layer kernel K8 (rqs_resnet_kernel.hpp; design notes in rqs_resnet.hip) for the bin counts
// other than 8 and 10 and for the block activations other than ReLU (round 4): a translation unit of their own.
#include "rqs_resnet_kernel.hpp"

namespace nfa {

#define NFA_K8_PICK(KB_, ACT_)                                                                                                      \
    (init_ks == 4 ? (inverse ? rqs_resnet_kernel<true, 1, 4, 0, KB_, false, ACT_> : rqs_resnet_kernel<false, 1, 4, 0, KB_, false, ACT_>) \
                  : (inverse ? rqs_resnet_kernel<true, 1, 2, 0, KB_, false, ACT_> : rqs_resnet_kernel<false, 1, 2, 0, KB_, false, ACT_>))

ResnetKernelFn resnet_bins_kernel(int K, bool inverse, int init_ks) {
    switch (K) {
        case 2: return NFA_K8_PICK(2, kActRelu);
        case 3: return NFA_K8_PICK(3, kActRelu);
        case 4: return NFA_K8_PICK(4, kActRelu);
        case 5: return NFA_K8_PICK(5, kActRelu);
        case 6: return NFA_K8_PICK(6, kActRelu);
        case 7: return NFA_K8_PICK(7, kActRelu);
        case 9: return NFA_K8_PICK(9, kActRelu);
        case 11: return NFA_K8_PICK(11, kActRelu);
        case 12: return NFA_K8_PICK(12, kActRelu);
        case 13: return NFA_K8_PICK(13, kActRelu);
        case 14: return NFA_K8_PICK(14, kActRelu);
        case 15: return NFA_K8_PICK(15, kActRelu);
        case 16: return NFA_K8_PICK(16, kActRelu);
        case 20: return NFA_K8_PICK(20, kActRelu);
        case 24: return NFA_K8_PICK(24, kActRelu);
        case 32: return NFA_K8_PICK(32, kActRelu);
    }
    return nullptr;
}

ResnetKernelFn resnet_activation_kernel(int activation, int K, bool inverse, int init_ks) {
    if (K != 8 && K != 10) return nullptr;
    switch (activation) {
        case NFA_ACTIVATION_LEAKY_RELU: return K == 8 ? NFA_K8_PICK(8, kActLeakyRelu) : NFA_K8_PICK(10, kActLeakyRelu);
        case NFA_ACTIVATION_ELU: return K == 8 ? NFA_K8_PICK(8, kActElu) : NFA_K8_PICK(10, kActElu);
        case NFA_ACTIVATION_TANH: return K == 8 ? NFA_K8_PICK(8, kActTanh) : NFA_K8_PICK(10, kActTanh);
    }
    return nullptr;
}

}  // namespace nfa
