// Thin torch cpp_extension binding over the SAME extern "C" entries of include/mil_hip.h (BASELINE north_star: "hand-written
// HIP C++ kernels bound through a thin torch cpp_extension C-ABI").  The reference has no FFI at all (its boundary is the
// Python class contract, model/utils.py:6-11); ctypes (_lib.py) binds every entry, and this extension binds the entries
// the token-side chain of the fusion step calls ~100 times per step, where a ctypes call (14 boxed arguments) costs more
// host time than the launch: tensors come in as at::Tensor, the stream as the integer torch.cuda.current_stream().cuda_stream.
// No kernels, no HIP headers, no torch types inside libmil_hip.so; built by __graft_entry__.build() into
// csrc/shim_build/mil_torch_shim.so.  Without it ops.py uses ctypes - the same C functions either way.
#include <torch/extension.h>
#include <cstdint>

#include "../../include/mil_hip.h"

using at::Tensor;
using c10::optional;

static inline const float* fp(const Tensor& t) { return t.data_ptr<float>(); }
static inline float* fpm(const Tensor& t) { return t.data_ptr<float>(); }
static inline const float* fpo(const optional<Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
static inline float* fpmo(const optional<Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
static inline int ld(const optional<Tensor>& t) { return t.has_value() ? (int)t->stride(0) : 0; }
static inline void* st(int64_t s) { return reinterpret_cast<void*>(s); }
static inline void ok(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, ": error ", rc, " (MIL_EINVAL -22 / MIL_ENOSPC -28 / hipError > 0)");
}

// mil_linear_small_fwd: y = act(x W^T + b) (+ residual) for <= 64 rows (aggregator.py:44-68, sam/common.py:21-26)
static void linear_small_fwd(const Tensor& x, const Tensor& W, const optional<Tensor>& b, int64_t act,
                             const optional<Tensor>& res, const Tensor& y, int64_t stream) {
    ok(mil_linear_small_fwd(fp(x), (int)x.stride(0), fp(W), (int)W.stride(0), fpo(b), (int)act, fpo(res), ld(res), fpm(y),
                            (int)y.stride(0), (int)x.size(0), (int)W.size(0), (int)x.size(1), st(stream)),
       "mil_linear_small_fwd");
}

// mil_linear_small_bwd: dx / dW / db of that layer in one launch (any of them absent)
static void linear_small_bwd(const Tensor& dy, const optional<Tensor>& yv, int64_t act, const Tensor& x, const Tensor& W,
                             const optional<Tensor>& dx, const optional<Tensor>& dW, const optional<Tensor>& db,
                             int64_t stream) {
    const int K = (int)x.size(1);
    ok(mil_linear_small_bwd(fp(dy), (int)dy.stride(0), fpo(yv), ld(yv), (int)act, fp(x), (int)x.stride(0), fp(W),
                            (int)W.stride(0), fpmo(dx), K, fpmo(dW), K, fpmo(db), (int)x.size(0), (int)W.size(0), K,
                            st(stream)),
       "mil_linear_small_bwd");
}

// mil_layernorm_fwd / mil_layernorm_bwd_res (sam/transformer.py:289-309 norm1..norm4, clip/model.py:153-159)
static void layernorm_fwd(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps, const Tensor& y,
                          const Tensor& stats, int64_t stream) {
    ok(mil_layernorm_fwd(fp(x), fp(gamma), fp(beta), (int)x.size(0), (int)x.size(1), (float)eps, fpm(y), fpm(stats),
                         st(stream)),
       "mil_layernorm_fwd");
}
static void layernorm_bwd_res(const Tensor& x, const Tensor& gamma, const Tensor& dy, const Tensor& stats,
                              const optional<Tensor>& dres, const Tensor& dx, const optional<Tensor>& dg,
                              const optional<Tensor>& db, const optional<Tensor>& ws, int64_t stream) {
    ok(mil_layernorm_bwd_res(fp(x), fp(gamma), fp(dy), fp(stats), fpo(dres), (int)x.size(0), (int)x.size(1), fpm(dx),
                             fpmo(dg), fpmo(db), fpmo(ws), st(stream)),
       "mil_layernorm_bwd_res");
}

// absorbed one-token attention, token side (csrc/absorbed_attn.hip)
static void absorb_query(const Tensor& qp, const Tensor& Wk, int64_t H, const Tensor& Qp, int64_t stream) {
    const int B = (int)qp.size(0), I = (int)qp.size(1), E = (int)Wk.size(1);
    ok(mil_absorb_query(fp(qp), fp(Wk), B, (int)H, I / (int)H, E, fpm(Qp), st(stream)), "mil_absorb_query");
}
static void absorb_query_bwd(const Tensor& qp, const Tensor& Wk, const Tensor& dQp, int64_t H, const optional<Tensor>& dqp,
                             const optional<Tensor>& dWk, int64_t stream) {
    const int B = (int)qp.size(0), I = (int)qp.size(1), E = (int)Wk.size(1);
    ok(mil_absorb_query_bwd(fp(qp), fp(Wk), fp(dQp), B, (int)H, I / (int)H, E, fpmo(dqp), fpmo(dWk), st(stream)),
       "mil_absorb_query_bwd");
}
static void value_proj_bwd(const Tensor& dO, const Tensor& Wv, const Tensor& pooled, const Tensor& dpooled, const Tensor& dWv,
                           const optional<Tensor>& dbv, int64_t stream) {
    const int B = (int)pooled.size(0), H = (int)pooled.size(1), E = (int)pooled.size(2), I = (int)Wv.size(0);
    ok(mil_value_proj_bwd(fp(dO), fp(Wv), fp(pooled), B, H, I / H, E, fpm(dpooled), fpm(dWv), fpmo(dbv), st(stream)),
       "mil_value_proj_bwd");
}

static int64_t abi_version() { return mil_abi_version(); }

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "torch cpp_extension binding of libmil_hip.so's token-side entries (include/mil_hip.h)";
    m.def("abi_version", &abi_version);
    m.def("linear_small_fwd", &linear_small_fwd);
    m.def("linear_small_bwd", &linear_small_bwd);
    m.def("layernorm_fwd", &layernorm_fwd);
    m.def("layernorm_bwd_res", &layernorm_bwd_res);
    m.def("absorb_query", &absorb_query);
    m.def("absorb_query_bwd", &absorb_query_bwd);
    m.def("value_proj_bwd", &value_proj_bwd);
}
