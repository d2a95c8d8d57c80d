// Instances of the whole-layer kernel K8 (rqs_resnet_kernel.hpp; design notes in rqs_resnet.hip) for conditioners WITH A
// CONTEXT (resnet.py:9-52, :92-100) beyond 8 / 10 bins with ReLU (round 5): the plain loop with the bin counts 2 .. 7, 9,
// 11 .. 16, 20, 24, 32 (ReLU) and with leaky ReLU / ELU / tanh blocks (8 / 10 bins).  Engine of its own and second pass
// behind K8h's instances of rqs_resnet_f16_ctx_{a,b}.hip.
#include "rqs_resnet_kernel.hpp"

namespace nfa {

#define NFA_K8_CTX_PICK(KB_, ACT_)                                                                                                    \
    (init_ks == 4 ? (inverse ? rqs_resnet_kernel<true, 1, 4, 0, KB_, true, ACT_> : rqs_resnet_kernel<false, 1, 4, 0, KB_, true, ACT_>) \
                  : (inverse ? rqs_resnet_kernel<true, 1, 2, 0, KB_, true, ACT_> : rqs_resnet_kernel<false, 1, 2, 0, KB_, true, ACT_>))

ResnetKernelFn resnet_context_kernel(int K, int activation, bool inverse, int init_ks) {
    switch (activation) {
        case NFA_ACTIVATION_RELU:
            switch (K) {
                case 2: return NFA_K8_CTX_PICK(2, kActRelu);
                case 3: return NFA_K8_CTX_PICK(3, kActRelu);
                case 4: return NFA_K8_CTX_PICK(4, kActRelu);
                case 5: return NFA_K8_CTX_PICK(5, kActRelu);
                case 6: return NFA_K8_CTX_PICK(6, kActRelu);
                case 7: return NFA_K8_CTX_PICK(7, kActRelu);
                case 9: return NFA_K8_CTX_PICK(9, kActRelu);
                case 11: return NFA_K8_CTX_PICK(11, kActRelu);
                case 12: return NFA_K8_CTX_PICK(12, kActRelu);
                case 13: return NFA_K8_CTX_PICK(13, kActRelu);
                case 14: return NFA_K8_CTX_PICK(14, kActRelu);
                case 15: return NFA_K8_CTX_PICK(15, kActRelu);
                case 16: return NFA_K8_CTX_PICK(16, kActRelu);
                case 20: return NFA_K8_CTX_PICK(20, kActRelu);
                case 24: return NFA_K8_CTX_PICK(24, kActRelu);
                case 32: return NFA_K8_CTX_PICK(32, kActRelu);
            }
            return nullptr;
        case NFA_ACTIVATION_LEAKY_RELU: return K == 8 ? NFA_K8_CTX_PICK(8, kActLeakyRelu) : K == 10 ? NFA_K8_CTX_PICK(10, kActLeakyRelu) : nullptr;
        case NFA_ACTIVATION_ELU: return K == 8 ? NFA_K8_CTX_PICK(8, kActElu) : K == 10 ? NFA_K8_CTX_PICK(10, kActElu) : nullptr;
        case NFA_ACTIVATION_TANH: return K == 8 ? NFA_K8_CTX_PICK(8, kActTanh) : K == 10 ? NFA_K8_CTX_PICK(10, kActTanh) : nullptr;
    }
    return nullptr;
}

}  // namespace nfa
