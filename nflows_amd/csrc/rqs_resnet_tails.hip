// Instances of the whole-layer kernel K8 (rqs_resnet_kernel.hpp; design notes in rqs_resnet.hip) for tails=None couplings
// (round 6; coupling.py:565-570 -> rational_quadratic.py:66-181 on [left, right] x [bottom, top], K + 1 derivative logits per
// feature): the plain loop on K1's register evaluator at every whole-layer bin count, ReLU blocks, no context.  A translation
// unit of its own.
#include "rqs_resnet_kernel.hpp"

namespace nfa {

#define NFA_K8_PICK_TAILS(KB_)                                                                                        \
    (init_ks == 4 ? (inverse ? rqs_resnet_kernel<true, 1, 4, 0, KB_, false, kActRelu, false, false>                   \
                             : rqs_resnet_kernel<false, 1, 4, 0, KB_, false, kActRelu, false, false>)                 \
                  : (inverse ? rqs_resnet_kernel<true, 1, 2, 0, KB_, false, kActRelu, false, false>                   \
                             : rqs_resnet_kernel<false, 1, 2, 0, KB_, false, kActRelu, false, false>))

ResnetKernelFn resnet_tails_kernel(int K, bool inverse, int init_ks) {
    switch (K) {
        case 2: return NFA_K8_PICK_TAILS(2);
        case 3: return NFA_K8_PICK_TAILS(3);
        case 4: return NFA_K8_PICK_TAILS(4);
        case 5: return NFA_K8_PICK_TAILS(5);
        case 6: return NFA_K8_PICK_TAILS(6);
        case 7: return NFA_K8_PICK_TAILS(7);
        case 8: return NFA_K8_PICK_TAILS(8);
        case 9: return NFA_K8_PICK_TAILS(9);
        case 10: return NFA_K8_PICK_TAILS(10);
        case 11: return NFA_K8_PICK_TAILS(11);
        case 12: return NFA_K8_PICK_TAILS(12);
        case 13: return NFA_K8_PICK_TAILS(13);
        case 14: return NFA_K8_PICK_TAILS(14);
        case 15: return NFA_K8_PICK_TAILS(15);
        case 16: return NFA_K8_PICK_TAILS(16);
        case 20: return NFA_K8_PICK_TAILS(20);
        case 24: return NFA_K8_PICK_TAILS(24);
        case 32: return NFA_K8_PICK_TAILS(32);
    }
    return nullptr;
}

}  // namespace nfa
