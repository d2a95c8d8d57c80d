// Instances of the whole-layer kernel K8h for conditioners WITH A CONTEXT, second half (see rqs_resnet_f16_ctx_a.hip):
// ReLU with 13 .. 16, 20, 24, 32 bins; leaky ReLU / ELU / tanh blocks with 8 / 10 bins.
#include "rqs_resnet_f16_kernel.hpp"

namespace nfa {
namespace k8h {

#define NFA_K8H_CTX_PICK(KB_, ACT_)                                                                  \
    (waves == 8 ? (inverse ? rqs_resnet_f16_kernel<true, 4, 8, KB_, true, kRing, ACT_>              \
                           : rqs_resnet_f16_kernel<false, 4, 8, KB_, true, kRing, ACT_>)             \
                : (inverse ? rqs_resnet_f16_kernel<true, 4, 4, KB_, true, kRing, ACT_>              \
                           : rqs_resnet_f16_kernel<false, 4, 4, KB_, true, kRing, ACT_>))

KernelFn context_kernel_b(int K, int activation, bool inverse, int waves) {
    switch (activation) {
        case NFA_ACTIVATION_RELU:
            switch (K) {
                case 13: return NFA_K8H_CTX_PICK(13, kActRelu);
                case 14: return NFA_K8H_CTX_PICK(14, kActRelu);
                case 15: return NFA_K8H_CTX_PICK(15, kActRelu);
                case 16: return NFA_K8H_CTX_PICK(16, kActRelu);
                case 20: return NFA_K8H_CTX_PICK(20, kActRelu);
                case 24: return NFA_K8H_CTX_PICK(24, kActRelu);
                case 32: return NFA_K8H_CTX_PICK(32, kActRelu);
            }
            return nullptr;
        case NFA_ACTIVATION_LEAKY_RELU: return K == 8 ? NFA_K8H_CTX_PICK(8, kActLeakyRelu) : K == 10 ? NFA_K8H_CTX_PICK(10, kActLeakyRelu) : nullptr;
        case NFA_ACTIVATION_ELU: return K == 8 ? NFA_K8H_CTX_PICK(8, kActElu) : K == 10 ? NFA_K8H_CTX_PICK(10, kActElu) : nullptr;
        case NFA_ACTIVATION_TANH: return K == 8 ? NFA_K8H_CTX_PICK(8, kActTanh) : K == 10 ? NFA_K8H_CTX_PICK(10, kActTanh) : nullptr;
    }
    return nullptr;
}

}  // namespace k8h
}  // namespace nfa
