// Per-element arithmetic of the affine layers (K2 `affine_coupling_kernel`, K2b `affine_ar_kernel`, K11
// `affine_mlp_kernel`): the scale activations and the map x -> x * scale + shift with its log-derivative.  No
// wave-level operation and no HIP type in here: the CPU suite compiles this file for the host
// (tests/test_affine_host.py) and holds it to the reference's vectors.
#pragma once
#include <math.h>

namespace nfa {

__device__ __forceinline__ float softplus1(float x) {
#pragma clang fp contract(off)
    return x > 20.0f ? x : log1pf(expf(x));
}

// scale activations, coupling.py:224-225 / autoregressive.py:101
__device__ __forceinline__ float scale_of(float u, int activation) {
#pragma clang fp contract(off)
    if (activation == NFA_SCALE_DEFAULT) {
        const float v = u + 2.0f;
        return 1.0f / (1.0f + expf(-v)) + 1e-3f;
    } else if (activation == NFA_SCALE_GENERAL) {
        float s = softplus1(u) + 1e-3f;
        s = s < 0.0f ? 0.0f : s;  // clamp(0, 3); NaN propagates like aten's clamp
        s = s > 3.0f ? 3.0f : s;
        return s;
    } else {  // NFA_SCALE_SOFTPLUS
        return softplus1(u) + 1e-3f;
    }
}



// one element of an affine layer given its scale: coupling.py:242-252 (forward y = x * scale + shift, logabsdet +=
// log scale; inverse x = (y - shift) / scale, logabsdet -= log scale), autoregressive.py:96-121
template <bool INVERSE>
__device__ __forceinline__ void affine_element(float xin, float shift, float sc, float& y, float& l) {
#pragma clang fp contract(off)
    const float ls = logf(sc);
    if (INVERSE) {
        y = (xin - shift) / sc;
        l = -ls;
    } else {
        y = xin * sc + shift;
        l = ls;
    }
}

}  // namespace nfa
