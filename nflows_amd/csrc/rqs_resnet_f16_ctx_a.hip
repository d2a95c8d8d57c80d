// Instances of the whole-layer kernel K8h (rqs_resnet_f16_kernel.hpp; design notes in rqs_resnet_f16.hip) for conditioners
// WITH A CONTEXT (resnet.py:9-52, :92-100) at the bin counts that have no loop of their own -- 2 .. 7, 9, 11, 12 here, the
// rest and the other activations in rqs_resnet_f16_ctx_b.hip (round 5; 8 and 10 bins with ReLU: rqs_resnet_f16.hip).
#include "rqs_resnet_f16_kernel.hpp"

namespace nfa {
namespace k8h {

#define NFA_K8H_CTX_PICK(KB_, ACT_)                                                                  \
    (waves == 8 ? (inverse ? rqs_resnet_f16_kernel<true, 4, 8, KB_, true, kRing, ACT_>              \
                           : rqs_resnet_f16_kernel<false, 4, 8, KB_, true, kRing, ACT_>)             \
                : (inverse ? rqs_resnet_f16_kernel<true, 4, 4, KB_, true, kRing, ACT_>              \
                           : rqs_resnet_f16_kernel<false, 4, 4, KB_, true, kRing, ACT_>))

KernelFn context_kernel_a(int K, int activation, bool inverse, int waves) {
    if (activation != NFA_ACTIVATION_RELU) return nullptr;
    switch (K) {
        case 2: return NFA_K8H_CTX_PICK(2, kActRelu);
        case 3: return NFA_K8H_CTX_PICK(3, kActRelu);
        case 4: return NFA_K8H_CTX_PICK(4, kActRelu);
        case 5: return NFA_K8H_CTX_PICK(5, kActRelu);
        case 6: return NFA_K8H_CTX_PICK(6, kActRelu);
        case 7: return NFA_K8H_CTX_PICK(7, kActRelu);
        case 9: return NFA_K8H_CTX_PICK(9, kActRelu);
        case 11: return NFA_K8H_CTX_PICK(11, kActRelu);
        case 12: return NFA_K8H_CTX_PICK(12, kActRelu);
    }
    return nullptr;
}

}  // namespace k8h
}  // namespace nfa
