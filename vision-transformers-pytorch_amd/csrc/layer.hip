// One C-ABI call per transformer layer (round 3).
//
// A Swin-S train step is ~500 kernel launches; enqueueing them one vtx_* call at a time costs the Python host 13.9 ms
// per step against 17.9 ms of GPU time (tools/probe/host_time.py) -- ~19 ctypes calls and ~28 tensor allocations per layer.
// vtx_layer_fwd / vtx_layer_bwd enqueue the SAME launches in the SAME order (they call the entry points of this
// library), from one descriptor: the host allocates one activation buffer per layer and passes addresses.  Results are
// bit-identical to the call-by-call path (tests/test_gpu_dispatch.py).
//
// Reference: the block  x1 = x + s1 * proj(attn(qkv(LN1 x)));  y = x1 + s2 * fc2(silu(fc1(LN2 x1)))  of
// models/vit.py:59-63 and models/swin_transformer.py:193-197 (window attention: :103-160), DropPath models/layer.py:172-180.
#include <hip/hip_runtime.h>

#include <vector>

#include "gemm_common.h"
#include "../../include/vtx.h"    // (after the internal header: its VTX_* macros shadow the internal enum of the same values)

namespace {

struct EventRing {                       // fork events for the side stream (never synchronised on the host)
  hipEvent_t ev[64];
  int n = 0, next = 0;
  hipEvent_t get() {
    if (n < 64) {
      if (hipEventCreateWithFlags(&ev[n], hipEventDisableTiming) != hipSuccess) return nullptr;
      return ev[n++];
    }
    hipEvent_t e = ev[next];
    next = (next + 1) & 63;
    return e;
  }
};
thread_local EventRing g_events;

// Per-launch HIP-event timing of the launches a vtx_layer_* call enqueues (bench.py's event-sampled steps: the roofline
// block must describe the kernels of the TIMED path -- one call per layer, compacted branches -- not a call-by-call replay of
// it).  Between vtx_timer_start() and vtx_timer_stop() every launch of a layer call is bracketed by two events on the stream
// it is enqueued on; vtx_timer_stop() synchronises and returns (what was launched, milliseconds) records.
struct TimerEntry { int tag, n, k, flags; int64_t rows; hipEvent_t e0, e1; };
std::vector<TimerEntry>* g_timer = nullptr;
int g_timer_base = 0;                     // dtype / head-dim bits of the layer call being recorded
struct TimerScope {
  TimerEntry e; hipStream_t st; bool on;
  TimerScope(int tag, int64_t rows, int n, int k, int flags, void* stream) : st((hipStream_t)stream), on(g_timer != nullptr) {
    if (!on) return;
    e.tag = tag; e.rows = rows; e.n = n; e.k = k; e.flags = flags | g_timer_base;
    on = hipEventCreate(&e.e0) == hipSuccess && hipEventCreate(&e.e1) == hipSuccess && hipEventRecord(e.e0, st) == hipSuccess;
  }
  ~TimerScope() {
    if (on && hipEventRecord(e.e1, st) == hipSuccess) g_timer->push_back(e);
  }
};
#define TCALL(tag, rows, n, k, flags, st, expr) ([&]() -> int { TimerScope ts_(tag, rows, n, k, flags, st); return (expr); }())
enum { F_RESID = 1, F_AUXOUT = 2, F_AUXIN = 4, F_MAPPED = 8, F_MASKED = 16, F_BF16 = 32, F_TWINS = 64 };   // | head dim << 8 (attention)

// dx = epi(dy @ W): the LDS-DMA kernel on the transposed weight copy where it applies (K = out-features, N =
// in-features), else the register-staged NN kernel on W itself -- the rule of vtx.functional.dgrad
int layer_dgrad(int dtype, const void* dy, const void* w, const void* wt, void* dx, int64_t M, int n_in, int k_out,
                const void* resid, const float* rowscale, int rps, const void* aux_in, int act, void* st) {
  if (dtype == VTX_BF16 && wt != nullptr && gemm_glds_ok(n_in, k_out))
    return vtx_gemm(0, dtype, dy, wt, dx, (int)M, n_in, k_out, k_out, k_out, n_in, nullptr, resid, rowscale, rps, nullptr,
                    aux_in, act, st);
  return vtx_gemm(1, dtype, dy, w, dx, (int)M, n_in, k_out, k_out, n_in, n_in, nullptr, resid, rowscale, rps, nullptr,
                  aux_in, act, st);
}

// One GEMM of a compacted branch (GemmArgs::perm): C[rows of perm order] = epi(A W^T) on the wave-private LDS-DMA kernel
int layer_gemm_mapped(const void* A, const void* W, void* Cc, int M, int Mk, int N, int K, const float* bias, const void* resid,
                      const float* rowscale, int T, void* aux_out, const void* aux_in, int act, const int* perm, void* st) {
  GemmArgs a;
  a.A = A; a.B = W; a.C = Cc; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.bias = bias; a.resid = resid; a.rowscale = rowscale; a.rows_per_scale = T; a.aux_out = aux_out; a.aux_in = aux_in;
  a.act = act; a.kscale = nullptr; a.k_per_scale = 1; a.kscale_const = 0.f; a.kchunk = ((K + 127) / 128) * 128;
  a.ksum_out = nullptr; a.perm = perm; a.map_T = T; a.Mk = Mk; a.map_magic = vtx_div_magic(T);
  if (!A || !W || !Cc || !perm) return VTX_ERR_NULL;
  if ((act == 2 || act == 4) && !aux_in) return VTX_ERR_NULL;
  return gemm_glds_launch_mapped(a, (hipStream_t)st);
}

}  // namespace

extern "C" {

/* sizeof the descriptors (0: VtxLayerFwd, 1: VtxLayerBwd): the bindings check their mirror structures against it */
int vtx_layer_desc_bytes(int which) {
  switch (which) {
    case 0: return (int)sizeof(VtxLayerFwd);
    case 1: return (int)sizeof(VtxLayerBwd);
    case 2: return (int)sizeof(VtxSrLayerFwd);
    case 3: return (int)sizeof(VtxSrLayerBwd);
    default: return 0;
  }
}

int vtx_timer_start(void) {
  if (g_timer == nullptr) g_timer = new std::vector<TimerEntry>();
  g_timer->clear();
  return VTX_OK;
}
/* -> number of records written (<= cap); synchronises on the recorded events, frees them and switches the timer off */
int vtx_timer_stop(VtxTimerRec* out, int cap) {
  if (g_timer == nullptr) return 0;
  int n = 0;
  for (TimerEntry& e : *g_timer) {
    float ms = 0.f;
    if (hipEventSynchronize(e.e1) == hipSuccess && hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess && out && n < cap) {
      out[n].tag = e.tag; out[n].n = e.n; out[n].k = e.k; out[n].flags = e.flags; out[n].rows = e.rows; out[n].ms = ms;
      ++n;
    }
    hipEventDestroy(e.e0); hipEventDestroy(e.e1);
  }
  delete g_timer;
  g_timer = nullptr;
  return n;
}

int vtx_layer_fwd(const VtxLayerFwd* a, void* stream) {
  if (!a || !a->x || !a->y || !a->ln1 || !a->qkv || !a->o || !a->x1 || !a->ln2 || !a->h) return VTX_ERR_NULL;
  if (a->M <= 0 || a->M > 0x7fffffff || a->C <= 0 || a->ff <= 0 || a->nH <= 0) return VTX_ERR_SHAPE;
  const int dt = a->dtype, M = (int)a->M, C = a->C, ff = a->ff;
  g_timer_base = (dt == VTX_BF16 ? F_BF16 : 0) | ((C / a->nH) << 8);
  if (a->perm1 != nullptr || a->perm2 != nullptr) {
    // ---- stochastic-depth compaction: each branch runs over ITS kept samples only (perm?: kept first, Bk? of them); the
    //      rows of dropped samples pass through (x1 = x, y = x1: copy-only tiles of the residual GEMMs), their saved
    //      activations are never written and never read.  bf16, window attention, N % 128 == 0, K % 64 == 0.
    const int T = a->rows_per_scale;
    if (!a->perm1 || !a->perm2 || dt != VTX_BF16 || T <= 0 || M != a->B * T ||
        a->Bk1 <= 0 || a->Bk1 > a->B || a->Bk2 <= 0 || a->Bk2 > a->B)
      return VTX_ERR_SHAPE;
    const int M1 = a->Bk1 * T, M2 = a->Bk2 * T;
    int rc = TCALL(VTX_T_LN_FWD, M1, C, 0, F_MAPPED, stream, vtx_layernorm_fwd_mapped(a->x, a->ln1_w, a->ln1_b, a->ln1, a->mean1, a->rstd1, M1, C, a->eps, dt, a->perm1, T, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M1, 3 * C, C, F_MAPPED, stream, layer_gemm_mapped(a->ln1, a->wq, a->qkv, M1, M1, 3 * C, C, a->bq, nullptr, nullptr, T, nullptr, nullptr, 0, a->perm1, stream));
    if (rc) return rc;
    if (a->attn_kind == VTX_ATTN_WINDOW)
      rc = TCALL(VTX_T_WATTN_FWD, M1, a->nH, a->L, F_MAPPED | (a->region ? F_MASKED : 0), stream,
                 vtx_wattn_fwd_mapped(a->qkv, a->o, a->lse, a->rel_pos, a->pos, a->region, a->perm1, a->Bk1, a->L, a->nH, a->H, a->W,
                                      a->win, a->shift, dt, stream));
    else
      rc = TCALL(VTX_T_ATTN_FWD, M1, a->nH, a->L, F_MAPPED, stream,
                 vtx_attention_fwd_mapped(a->qkv, a->o, a->lse, a->perm1, a->Bk1, a->L, a->nH, C / a->nH, dt, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M1, C, C, F_MAPPED | F_RESID, stream, layer_gemm_mapped(a->o, a->wo, a->x1, M, M1, C, C, a->bo, a->x, a->s1, T, nullptr, nullptr, 0, a->perm1, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_LN_FWD, M2, C, 0, F_MAPPED, stream, vtx_layernorm_fwd_mapped(a->x1, a->ln2_w, a->ln2_b, a->ln2, a->mean2, a->rstd2, M2, C, a->eps, dt, a->perm2, T, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M2, ff, C, F_MAPPED | (a->z ? F_AUXOUT : 0), stream, layer_gemm_mapped(a->ln2, a->w1, a->h, M2, M2, ff, C, a->b1, nullptr, nullptr, T, a->z, nullptr, 1, a->perm2, stream));
    if (rc) return rc;
    return TCALL(VTX_T_GEMM, M2, C, ff, F_MAPPED | F_RESID, stream, layer_gemm_mapped(a->h, a->w2, a->y, M, M2, C, ff, a->b2, a->x1, a->s2, T, nullptr, nullptr, 0, a->perm2, stream));
  }
  int rc;
  if (ln_gemm_ok(dt, M, C, 3 * C)) {
    // (option LN_FOLD bit 3) narrow layers: norm_attn runs on the row operands of the qkv projection, ln1 / mean1 / rstd1 stored on the side
    rc = TCALL(VTX_T_LN_GEMM, M, 3 * C, C, 0, stream,
               ln_gemm_launch(a->x, a->ln1_w, a->ln1_b, a->eps, a->ln1, a->mean1, a->rstd1, a->wq, a->bq, a->qkv, M, C, 3 * C, (hipStream_t)stream));
    if (rc) return rc;
  } else {
    rc = TCALL(VTX_T_LN_FWD, M, C, 0, 0, stream, vtx_layernorm_fwd(a->x, a->ln1_w, a->ln1_b, a->ln1, a->mean1, a->rstd1, a->M, C, a->eps, dt, 0, 0, 0, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M, 3 * C, C, 0, stream, vtx_gemm(0, dt, a->ln1, a->wq, a->qkv, M, 3 * C, C, C, C, 3 * C, a->bq, nullptr, nullptr, 1, nullptr, nullptr, 0, stream));
    if (rc) return rc;
  }
  if (a->attn_kind == VTX_ATTN_WINDOW)
    rc = TCALL(VTX_T_WATTN_FWD, M, a->nH, a->L, a->region ? F_MASKED : 0, stream,
               vtx_wattn_fwd(a->qkv, a->o, a->lse, a->rel_pos, a->pos, a->region, a->B, a->L, a->nH, a->H, a->W, a->win, a->shift, dt, stream));
  else if (a->attn_kind == VTX_ATTN_GLOBAL)
    rc = TCALL(VTX_T_ATTN_FWD, M, a->nH, a->L, 0, stream,
               vtx_attention_fwd(a->qkv, a->o, a->lse, nullptr, nullptr, a->B, a->L, a->nH, C / a->nH, 0, 0, 0, 0, 0, dt, stream));
  else
    return VTX_ERR_SHAPE;
  if (rc) return rc;
  rc = TCALL(VTX_T_GEMM, M, C, C, F_RESID, stream, vtx_gemm(0, dt, a->o, a->wo, a->x1, M, C, C, C, C, C, a->bo, a->x, a->s1, a->rows_per_scale, nullptr, nullptr, 0, stream));
  if (rc) return rc;
  // (option LN_FOLD bit 2) ... with norm_ff on its row operands: x1 in (it is the residual anyway), ln2 / mean2 / rstd2 and y out
  if (mlp_fused_lnf_ok(dt, M, C, ff))
    return TCALL(VTX_T_MLP_FWD, M, C, ff, F_RESID | F_AUXOUT, stream,
                 mlp_fused_fwd_ln(a->x1, a->ln2_w, a->ln2_b, a->eps, a->ln2, a->mean2, a->rstd2, a->w1, a->b1, a->w2, a->b2, a->s2, a->rows_per_scale,
                                  a->y, M, C, ff, (hipStream_t)stream));
  rc = TCALL(VTX_T_LN_FWD, M, C, 0, 0, stream, vtx_layernorm_fwd(a->x1, a->ln2_w, a->ln2_b, a->ln2, a->mean2, a->rstd2, a->M, C, a->eps, dt, 0, 0, 0, stream));
  if (rc) return rc;
  // narrow stages (C = 64 / 96): the whole MLP in one launch, nothing ff-wide stored (the backward recomputes z and h: mlp_fused.hip)
  if (mlp_fused_ok(dt, M, C, ff))
    return TCALL(VTX_T_MLP_FWD, M, C, ff, F_RESID, stream,
                 mlp_fused_fwd(a->ln2, a->w1, a->b1, a->w2, a->b2, a->x1, a->s2, a->rows_per_scale, a->y, nullptr, nullptr, M, C, ff, (hipStream_t)stream));
  rc = TCALL(VTX_T_GEMM, M, ff, C, a->z ? F_AUXOUT : 0, stream, vtx_gemm(0, dt, a->ln2, a->w1, a->h, M, ff, C, C, C, ff, a->b1, nullptr, nullptr, 1, a->z, nullptr, 1, stream));
  if (rc) return rc;
  return TCALL(VTX_T_GEMM, M, C, ff, F_RESID, stream,
               vtx_gemm(0, dt, a->h, a->w2, a->y, M, C, ff, ff, ff, C, a->b2, a->x1, a->s2, a->rows_per_scale, nullptr, nullptr, 0, stream));
}

int vtx_layer_bwd(const VtxLayerBwd* a, void* stream, void* side_stream) {
  if (!a || !a->dy || !a->dx || !a->x || !a->z || !a->dz || !a->dln2 || !a->dx1 || !a->dout || !a->dqkv || !a->dln1)
    return VTX_ERR_NULL;
  if (a->M <= 0 || a->M > 0x7fffffff || a->C <= 0 || a->ff <= 0 || a->nH <= 0) return VTX_ERR_SHAPE;
  const int dt = a->dtype, C = a->C, ff = a->ff, rps = a->rows_per_scale;
  const int64_t M = a->M;
  const bool mapped = a->perm1 != nullptr || a->perm2 != nullptr;
  g_timer_base = (dt == VTX_BF16 ? F_BF16 : 0) | ((C / a->nH) << 8);
  if (mapped) {
    const int T = rps;
    if (!a->perm1 || !a->perm2 || dt != VTX_BF16 || T <= 0 || M != (int64_t)a->B * T ||
        a->Bk1 <= 0 || a->Bk1 > a->B || a->Bk2 <= 0 || a->Bk2 > a->B || !a->w2t || !a->w1t || !a->wot || !a->wqt ||
        !a->s1 || !a->s2)
      return VTX_ERR_SHAPE;
    const int M1 = a->Bk1 * T, M2 = a->Bk2 * T;
    int rc = TCALL(VTX_T_GEMM, M2, ff, C, F_MAPPED | F_AUXIN, stream, layer_gemm_mapped(a->dy, a->w2t, a->dz, M2, M2, ff, C, nullptr, nullptr, a->s2, T, nullptr, a->z, 2, a->perm2, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M2, C, ff, F_MAPPED, stream, layer_gemm_mapped(a->dz, a->w1t, a->dln2, M2, M2, C, ff, nullptr, nullptr, nullptr, T, nullptr, nullptr, 0, a->perm2, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_LN_BWD, M, C, 0, F_MAPPED, stream,
               vtx_layernorm_bwd_mapped(a->dln2, a->x1, a->mean2, a->rstd2, a->ln2_w, a->dy, a->dx1, a->ln2_ws, a->ln_ws_bytes, M, M2, C, dt,
                                        a->perm2, T, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M1, C, C, F_MAPPED, stream, layer_gemm_mapped(a->dx1, a->wot, a->dout, M1, M1, C, C, nullptr, nullptr, a->s1, T, nullptr, nullptr, 0, a->perm1, stream));
    if (rc) return rc;
    if (a->attn_kind == VTX_ATTN_WINDOW)
      rc = TCALL(VTX_T_WATTN_BWD, M1, a->nH, a->L, F_MAPPED | (a->region ? F_MASKED : 0), stream,
                 vtx_wattn_bwd_mapped(a->qkv, a->o, a->dout, a->lse, a->rel_pos, a->pos, a->region, a->dqkv, a->attn_ws, a->attn_ws_bytes,
                                      a->inv_cells, a->inv_count, a->perm1, a->Bk1, a->L, a->nH, a->H, a->W, a->win, a->shift, dt, stream));
    else
      rc = TCALL(VTX_T_ATTN_BWD, M1, a->nH, a->L, F_MAPPED, stream,
                 vtx_attention_bwd_mapped(a->qkv, a->o, a->dout, a->lse, a->dqkv, a->perm1, a->Bk1, a->L, a->nH, C / a->nH, dt, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M1, C, 3 * C, F_MAPPED, stream, layer_gemm_mapped(a->dqkv, a->wqt, a->dln1, M1, M1, C, 3 * C, nullptr, nullptr, nullptr, T, nullptr, nullptr, 0, a->perm1, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_LN_BWD, M, C, 0, F_MAPPED, stream,
               vtx_layernorm_bwd_mapped(a->dln1, a->x, a->mean1, a->rstd1, a->ln1_w, a->dx1, a->dx, a->ln1_ws, a->ln_ws_bytes, M, M1, C, dt,
                                        a->perm1, T, stream));
    if (rc) return rc;
  }
  if (!mapped) {
    // ---- MLP branch
    int rc;
    bool ln2_folded = false;
    if (mlp_fused_ok(dt, M, C, ff)) {
      // (the forward kept neither z nor h: both are recomputed; h lands in the layer's own h buffer for the weight gradient below)
      if (!a->h || !a->b1) return VTX_ERR_NULL;
      if (mlp_fused_ln_ok(dt, M, C, ff)) {
        // (option LN_FOLD bit 0) ... and the LayerNorm backward of norm_ff in its epilogue: dx1 out, dln2 never stored, one launch less
        if (a->ln_ws_bytes < vtx_layernorm_bwd_workspace(M, C)) return VTX_ERR_WORKSPACE;
        rc = TCALL(VTX_T_MLP_BWD, M, C, ff, F_RESID, stream,
                   mlp_fused_bwd_ln(a->ln2, a->dy, a->w1, a->b1, a->w2, a->s2, rps, const_cast<void*>(a->h), a->dz, a->x1, a->mean2, a->rstd2,
                                    a->ln2_w, a->dx1, (float*)a->ln2_ws, vtx_layernorm_bwd_blocks(M, C), M, C, ff, (hipStream_t)stream));
        if (rc) return rc;
        ln2_folded = true;
      } else {
        rc = TCALL(VTX_T_MLP_BWD, M, C, ff, 0, stream,
                   mlp_fused_bwd(a->ln2, a->dy, a->w1, a->b1, a->w2, a->s2, rps, const_cast<void*>(a->h), a->dz, a->dln2, M, C, ff, (hipStream_t)stream));
        if (rc) return rc;
      }
    } else {
      rc = TCALL(VTX_T_GEMM, M, ff, C, F_AUXIN, stream, layer_dgrad(dt, a->dy, a->w2, a->w2t, a->dz, M, ff, C, nullptr, a->s2, rps, a->z, 2, stream));
      if (rc) return rc;
      rc = TCALL(VTX_T_GEMM, M, C, ff, 0, stream, layer_dgrad(dt, a->dz, a->w1, a->w1t, a->dln2, M, C, ff, nullptr, nullptr, 1, nullptr, 0, stream));
      if (rc) return rc;
    }
    if (!ln2_folded) {
      rc = TCALL(VTX_T_LN_BWD, M, C, 0, 0, stream,
                 vtx_layernorm_bwd(a->dln2, a->x1, a->mean2, a->rstd2, a->ln2_w, a->dy, a->dx1, nullptr, nullptr, a->ln2_ws, a->ln_ws_bytes,
                                   M, C, dt, 0, 0, 0, stream));
      if (rc) return rc;
    }
    // ---- attention branch
    rc = TCALL(VTX_T_GEMM, M, C, C, 0, stream, layer_dgrad(dt, a->dx1, a->wo, a->wot, a->dout, M, C, C, nullptr, a->s1, rps, nullptr, 0, stream));
    if (rc) return rc;
    if (a->attn_kind == VTX_ATTN_WINDOW)
      rc = TCALL(VTX_T_WATTN_BWD, M, a->nH, a->L, a->region ? F_MASKED : 0, stream,
                 vtx_wattn_bwd(a->qkv, a->o, a->dout, a->lse, a->rel_pos, a->pos, a->region, a->dqkv, nullptr, a->attn_ws, a->attn_ws_bytes,
                               a->inv_cells, a->inv_count, a->B, a->L, a->nH, a->H, a->W, a->win, a->shift, dt, stream));
    else if (a->attn_kind == VTX_ATTN_GLOBAL)
      rc = TCALL(VTX_T_ATTN_BWD, M, a->nH, a->L, 0, stream,
                 vtx_attention_bwd(a->qkv, a->o, a->dout, a->lse, nullptr, nullptr, nullptr, nullptr, a->dqkv, nullptr, 0, a->attn_ws,
                                   a->attn_ws_bytes, a->B, a->L, a->nH, C / a->nH, 0, 0, 0, 0, 0, dt, stream));
    else
      return VTX_ERR_SHAPE;
    if (rc) return rc;
    if (a->wqt != nullptr && dgrad_ln_ok(dt, M, C, 3 * C)) {
      // (option LN_FOLD bit 1) narrow layers: the qkv input gradient and the norm_attn backward in ONE launch, dln1 never stored
      if (a->ln_ws_bytes < vtx_layernorm_bwd_workspace(M, C)) return VTX_ERR_WORKSPACE;
      rc = TCALL(VTX_T_DGRAD_LN, M, C, 3 * C, 0, stream,
                 dgrad_ln_launch(a->dqkv, a->wqt, a->x, a->mean1, a->rstd1, a->ln1_w, a->dx1, a->dx, (float*)a->ln1_ws, vtx_layernorm_bwd_blocks(M, C),
                                 M, C, 3 * C, (hipStream_t)stream));
    } else {
      rc = TCALL(VTX_T_GEMM, M, C, 3 * C, 0, stream, layer_dgrad(dt, a->dqkv, a->wq, a->wqt, a->dln1, M, C, 3 * C, nullptr, nullptr, 1, nullptr, 0, stream));
      if (rc) return rc;
      rc = TCALL(VTX_T_LN_BWD, M, C, 0, 0, stream,
                 vtx_layernorm_bwd(a->dln1, a->x, a->mean1, a->rstd1, a->ln1_w, a->dx1, a->dx, nullptr, nullptr, a->ln1_ws, a->ln_ws_bytes,
                                   M, C, dt, 0, 0, 0, stream));
    }
    if (rc) return rc;
  }
  // ---- the four weight gradients + the layer's column reductions: one grouped launch + one reduce launch, on the side
  //      stream when given (fork: it waits for everything enqueued on `stream` so far; the caller joins once per backward)
  void* ws = stream;
  if (side_stream != nullptr && side_stream != stream) {
    hipEvent_t e = g_events.get();
    if (!e || hipEventRecord(e, (hipStream_t)stream) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)side_stream, e, 0) != hipSuccess)
      return VTX_ERR_LAUNCH;
    ws = side_stream;
  }
  const void* dys[4] = {a->dy, a->dz, a->dx1, a->dqkv};
  const void* xs[4] = {a->h, a->ln2, a->o, a->ln1};
  float* dWs[4] = {a->dW2, a->dW1, a->dWo, a->dWq};
  float* dbs[4] = {a->db2, a->db1, a->dbo, a->dbq};
  const int Ns[4] = {C, ff, C, 3 * C}, Ks[4] = {ff, C, C, C};
  const int64_t ldy[4] = {C, ff, C, 3 * C}, ldx[4] = {ff, C, C, C};
  // compacted: every problem contracts over the kept samples' tokens only (the rows of dropped samples were never written);
  // the DropPath constant goes onto the problems whose dy does not carry it yet (fc2: dy, proj: dx1), not onto dz / dqkv
  const float* rs[4] = {mapped ? nullptr : a->s2, nullptr, mapped ? nullptr : a->s1, nullptr};
  const int* perms[4] = {a->perm2, a->perm2, a->perm1, a->perm1};
  const int kept[4] = {a->Bk2 * rps, a->Bk2 * rps, a->Bk1 * rps, a->Bk1 * rps};
  const float scl[4] = {a->scale_const, 1.f, a->scale_const, 1.f};
  const int ncol = a->attn_kind == VTX_ATTN_WINDOW ? 3 : 2;
  const float* cpart[4] = {(const float*)a->ln2_ws, (const float*)a->ln1_ws, (const float*)a->attn_ws, nullptr};
  float* cout0[4] = {a->dg2, a->dg1, a->drel, nullptr};
  float* cout1[4] = {a->dbe2, a->dbe1, nullptr, nullptr};
  const int ntab = (2 * a->win - 1) * (2 * a->win - 1);
  const int cnb[4] = {vtx_layernorm_bwd_blocks(M, C), vtx_layernorm_bwd_blocks(M, C),
                      ncol == 3 ? vtx_wattn_bwd_parts(mapped ? a->Bk1 : a->B, a->nH, a->H, a->W, a->win) : 0, 0};
  const int cC[4] = {C, C, ncol == 3 ? ntab * a->nH : 0, 0};
  const int cld[4] = {2 * C, 2 * C, ncol == 3 ? vtx_wattn_bwd_part_ld(a->nH) : 0, 0};
  return TCALL(VTX_T_WGRAD, mapped ? (int64_t)(kept[0] > kept[2] ? kept[0] : kept[2]) : M, C, ff, mapped ? F_MAPPED : 0, ws,
               vtx_wgrad_group_mapped(dt, 4, dys, xs, dWs, dbs, Ns, Ks, ldy, ldx, rs, mapped ? perms : nullptr, kept, scl, rps,
                                      a->scale_const, M, a->wgrad_ws, a->wgrad_ws_bytes, ncol, cpart, cout0, cout1, cnb, cC, cld,
                                      a->accumulate, ws));
}

// ------------------------------------------------------------------------------------------------------------------
// One PVT block / the global half of a Twins-SVT layer per call (vtx.h VtxSrLayerFwd / Bwd): exactly the launches of
// vtx.functional.PvtLayerFn's call-by-call path, in its order, with its arguments.
int vtx_srlayer_fwd(const VtxSrLayerFwd* a, void* stream) {
  if (!a || !a->x || !a->y || !a->ln1 || !a->q || !a->kv || !a->o || !a->x1 || !a->ln2 || !a->h) return VTX_ERR_NULL;
  if (a->M <= 0 || a->M > 0x7fffffff || a->C <= 0 || a->ff <= 0 || a->nH <= 0 || a->B <= 0 || a->L <= 0 || a->Lk <= 0 || a->r < 1)
    return VTX_ERR_SHAPE;
  const int dt = a->dtype, M = (int)a->M, C = a->C, ff = a->ff, r = a->r;
  g_timer_base = (dt == VTX_BF16 ? F_BF16 : 0) | ((C / a->nH) << 8);
  int rc;
  if (ln_gemm_ok(dt, M, C, C)) {
    // (option LN_FOLD bit 3) narrow stages: norm_attn on the row operands of the q projection; ln1 (the reduction branch reads it) on the side
    rc = TCALL(VTX_T_LN_GEMM, M, C, C, 0, stream,
               ln_gemm_launch(a->x, a->ln1_w, a->ln1_b, a->eps, a->ln1, a->mean1, a->rstd1, a->wq, nullptr, a->q, M, C, C, (hipStream_t)stream));
    if (rc) return rc;
  } else {
    rc = TCALL(VTX_T_LN_FWD, M, C, 0, 0, stream, vtx_layernorm_fwd(a->x, a->ln1_w, a->ln1_b, a->ln1, a->mean1, a->rstd1, a->M, C, a->eps, dt, 0, 0, 0, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M, C, C, 0, stream, vtx_gemm(0, dt, a->ln1, a->wq, a->q, M, C, C, C, C, C, nullptr, nullptr, nullptr, 1, nullptr, nullptr, 0, stream));
    if (rc) return rc;
  }
  const void* kvin = a->ln1;
  const int rows = a->B * a->Lk;
  if (r > 1) {
    if (!a->patches || !a->red || !a->wsr) return VTX_ERR_NULL;
    const int K = r * r * C;
    if (a->twins) rc = TCALL(VTX_T_GATHER, rows, K, 0, F_TWINS, stream, vtx_twins_subsample_fwd(a->ln1, a->patches, a->splitk ? a->patches_t : nullptr, a->B, a->H, a->W, C, r, dt, stream));
    else rc = TCALL(VTX_T_GATHER, rows, K, 0, 0, stream, vtx_patchify_fwd(a->ln1, a->patches, a->B, a->H, a->W, C, r, a->skip, dt, stream));
    if (rc) return rc;
    if (a->splitk) {
      if (!a->twins || !a->patches_t || !a->wsr_t || !a->red32 || !a->splitk_ws) return VTX_ERR_NULL;
      rc = TCALL(VTX_T_WGRAD_SPLITK, K, rows, C, 0, stream,
                 vtx_wgrad(dt, a->patches_t, a->wsr_t, (float*)a->red32, nullptr, K, rows, C, rows, C, nullptr, 1, 0.f, a->splitk_ws,
                           a->splitk_ws_bytes, stream));
      if (rc) return rc;
      rc = vtx_bias_cast((const float*)a->red32, a->bsr, a->red, rows, C, dt, stream);
    } else {
      rc = TCALL(VTX_T_GEMM, rows, C, K, 0, stream, vtx_gemm(0, dt, a->patches, a->wsr, a->red, rows, C, K, K, K, C, a->bsr, nullptr, nullptr, 1, nullptr, nullptr, 0, stream));
    }
    if (rc) return rc;
    kvin = a->red;
    if (a->srn_w != nullptr) {
      if (!a->kvin || !a->means || !a->rstds) return VTX_ERR_NULL;
      rc = TCALL(VTX_T_LN_FWD, rows, C, 0, 0, stream, vtx_layernorm_fwd(a->red, a->srn_w, a->srn_b, a->kvin, a->means, a->rstds, rows, C, a->eps, dt, 0, 0, 0, stream));
      if (rc) return rc;
      kvin = a->kvin;
    }
  }
  rc = TCALL(VTX_T_GEMM, rows, 2 * C, C, 0, stream, vtx_gemm(0, dt, kvin, a->wkv, a->kv, rows, 2 * C, C, C, C, 2 * C, nullptr, nullptr, nullptr, 1, nullptr, nullptr, 0, stream));
  if (rc) return rc;
  rc = TCALL(VTX_T_SRATTN_FWD, M, a->nH, a->Lk, 0, stream, vtx_srattn_fwd(a->q, a->kv, a->o, a->lse, a->B, a->L, a->Lk, a->nH, C / a->nH, dt, stream));
  if (rc) return rc;
  rc = TCALL(VTX_T_GEMM, M, C, C, F_RESID, stream, vtx_gemm(0, dt, a->o, a->wo, a->x1, M, C, C, C, C, C, a->bo, a->x, a->s1, a->rows_per_scale, nullptr, nullptr, 0, stream));
  if (rc) return rc;
  if (mlp_fused_lnf_ok(dt, M, C, ff))             // (as in vtx_layer_fwd: norm_ff on the row operands of the fused MLP)
    return TCALL(VTX_T_MLP_FWD, M, C, ff, F_RESID | F_AUXOUT, stream,
                 mlp_fused_fwd_ln(a->x1, a->ln2_w, a->ln2_b, a->eps, a->ln2, a->mean2, a->rstd2, a->w1, a->b1, a->w2, a->b2, a->s2, a->rows_per_scale,
                                  a->y, M, C, ff, (hipStream_t)stream));
  rc = TCALL(VTX_T_LN_FWD, M, C, 0, 0, stream, vtx_layernorm_fwd(a->x1, a->ln2_w, a->ln2_b, a->ln2, a->mean2, a->rstd2, a->M, C, a->eps, dt, 0, 0, 0, stream));
  if (rc) return rc;
  if (mlp_fused_ok(dt, M, C, ff))                 // (as in vtx_layer_fwd: PVT-Small / Twins-SVT-S stage 1, C = 64)
    return TCALL(VTX_T_MLP_FWD, M, C, ff, F_RESID, stream,
                 mlp_fused_fwd(a->ln2, a->w1, a->b1, a->w2, a->b2, a->x1, a->s2, a->rows_per_scale, a->y, nullptr, nullptr, M, C, ff, (hipStream_t)stream));
  rc = TCALL(VTX_T_GEMM, M, ff, C, a->z ? F_AUXOUT : 0, stream, vtx_gemm(0, dt, a->ln2, a->w1, a->h, M, ff, C, C, C, ff, a->b1, nullptr, nullptr, 1, a->z, nullptr, 1, stream));
  if (rc) return rc;
  return TCALL(VTX_T_GEMM, M, C, ff, F_RESID, stream,
               vtx_gemm(0, dt, a->h, a->w2, a->y, M, C, ff, ff, ff, C, a->b2, a->x1, a->s2, a->rows_per_scale, nullptr, nullptr, 0, stream));
}

int vtx_srlayer_bwd(const VtxSrLayerBwd* a, void* stream, void* side_stream) {
  if (!a || !a->dy || !a->dx || !a->x || !a->z || !a->dz || !a->dln2 || !a->dx1 || !a->dout || !a->dq || !a->dkv || !a->dkvin || !a->dln1)
    return VTX_ERR_NULL;
  if (a->M <= 0 || a->M > 0x7fffffff || a->C <= 0 || a->ff <= 0 || a->nH <= 0 || a->r < 1) return VTX_ERR_SHAPE;
  const int dt = a->dtype, C = a->C, ff = a->ff, rps = a->rows_per_scale, r = a->r;
  const int64_t M = a->M;
  const int rows = a->B * a->Lk;
  g_timer_base = (dt == VTX_BF16 ? F_BF16 : 0) | ((C / a->nH) << 8);
  // ---- MLP branch
  int rc;
  bool ln2_folded = false;
  if (mlp_fused_ok(dt, M, C, ff)) {
    if (!a->h || !a->b1) return VTX_ERR_NULL;
    if (mlp_fused_ln_ok(dt, M, C, ff)) {               // (as in vtx_layer_bwd: the norm_ff backward in the epilogue of the fused-MLP backward)
      if (a->ln_ws_bytes < vtx_layernorm_bwd_workspace(M, C)) return VTX_ERR_WORKSPACE;
      rc = TCALL(VTX_T_MLP_BWD, M, C, ff, F_RESID, stream,
                 mlp_fused_bwd_ln(a->ln2, a->dy, a->w1, a->b1, a->w2, a->s2, rps, const_cast<void*>(a->h), a->dz, a->x1, a->mean2, a->rstd2,
                                  a->ln2_w, a->dx1, (float*)a->ln2_ws, vtx_layernorm_bwd_blocks(M, C), M, C, ff, (hipStream_t)stream));
      if (rc) return rc;
      ln2_folded = true;
    } else {
      rc = TCALL(VTX_T_MLP_BWD, M, C, ff, 0, stream,
                 mlp_fused_bwd(a->ln2, a->dy, a->w1, a->b1, a->w2, a->s2, rps, const_cast<void*>(a->h), a->dz, a->dln2, M, C, ff, (hipStream_t)stream));
      if (rc) return rc;
    }
  } else {
    rc = TCALL(VTX_T_GEMM, M, ff, C, F_AUXIN, stream, layer_dgrad(dt, a->dy, a->w2, a->w2t, a->dz, M, ff, C, nullptr, a->s2, rps, a->z, 2, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M, C, ff, 0, stream, layer_dgrad(dt, a->dz, a->w1, a->w1t, a->dln2, M, C, ff, nullptr, nullptr, 1, nullptr, 0, stream));
    if (rc) return rc;
  }
  if (!ln2_folded) {
    rc = TCALL(VTX_T_LN_BWD, M, C, 0, 0, stream,
               vtx_layernorm_bwd(a->dln2, a->x1, a->mean2, a->rstd2, a->ln2_w, a->dy, a->dx1, nullptr, nullptr, a->ln2_ws, a->ln_ws_bytes, M, C, dt, 0, 0, 0, stream));
    if (rc) return rc;
  }
  // ---- attention branch
  rc = TCALL(VTX_T_GEMM, M, C, C, 0, stream, layer_dgrad(dt, a->dx1, a->wo, a->wot, a->dout, M, C, C, nullptr, a->s1, rps, nullptr, 0, stream));
  if (rc) return rc;
  rc = TCALL(VTX_T_SRATTN_BWD, M, a->nH, a->Lk, 0, stream,
             vtx_srattn_bwd(a->q, a->kv, a->o, a->dout, a->lse, a->dq, a->dkv, a->attn_ws, a->attn_ws_bytes, a->B, a->L, a->Lk, a->nH, C / a->nH, dt, stream));
  if (rc) return rc;
  const bool srn = r > 1 && a->srn_w != nullptr;
  rc = TCALL(VTX_T_GEMM, rows, C, 2 * C, 0, stream, layer_dgrad(dt, a->dkv, a->wkv, a->wkvt, a->dkvin, rows, C, 2 * C, nullptr, nullptr, 1, nullptr, 0, stream));
  if (rc) return rc;
  if (r > 1) {
    if (!a->dpatches || !a->patches || !a->red) return VTX_ERR_NULL;
    const int K = r * r * C;
    const void* dred = a->dkvin;
    if (srn) {
      if (!a->dred || !a->lns_ws) return VTX_ERR_NULL;
      rc = TCALL(VTX_T_LN_BWD, rows, C, 0, 0, stream,
                 vtx_layernorm_bwd(a->dkvin, a->red, a->means, a->rstds, a->srn_w, nullptr, a->dred, nullptr, nullptr, a->lns_ws, a->lns_ws_bytes, rows, C, dt, 0, 0, 0, stream));
      if (rc) return rc;
      dred = a->dred;
    }
    rc = TCALL(VTX_T_GEMM, rows, K, C, 0, stream, layer_dgrad(dt, dred, a->wsr, a->wsrt, a->dpatches, rows, K, C, nullptr, nullptr, 1, nullptr, 0, stream));
    if (rc) return rc;
    rc = TCALL(VTX_T_GEMM, M, C, C, 0, stream, layer_dgrad(dt, a->dq, a->wq, a->wqt, a->dln1, M, C, C, nullptr, nullptr, 1, nullptr, 0, stream));
    if (rc) return rc;
    if (a->twins) rc = TCALL(VTX_T_GATHER, rows, K, 0, F_TWINS | 1, stream, vtx_twins_subsample_bwd(a->dpatches, a->dln1, a->B, a->H, a->W, C, r, 1, dt, stream));
    else rc = TCALL(VTX_T_GATHER, rows, K, 0, 1, stream, vtx_patchify_bwd(a->dpatches, a->dln1, a->B, a->H, a->W, C, r, a->skip, 1, dt, stream));
    if (rc) return rc;
  } else {
    rc = TCALL(VTX_T_GEMM, M, C, C, F_RESID, stream, layer_dgrad(dt, a->dq, a->wq, a->wqt, a->dln1, M, C, C, a->dkvin, nullptr, 1, nullptr, 0, stream));
    if (rc) return rc;
  }
  rc = TCALL(VTX_T_LN_BWD, M, C, 0, 0, stream,
             vtx_layernorm_bwd(a->dln1, a->x, a->mean1, a->rstd1, a->ln1_w, a->dx1, a->dx, nullptr, nullptr, a->ln1_ws, a->ln_ws_bytes, M, C, dt, 0, 0, 0, stream));
  if (rc) return rc;
  // ---- weight gradients: on the side stream when given (fork after everything enqueued on `stream` so far)
  void* ws = stream;
  if (side_stream != nullptr && side_stream != stream) {
    hipEvent_t e = g_events.get();
    if (!e || hipEventRecord(e, (hipStream_t)stream) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)side_stream, e, 0) != hipSuccess)
      return VTX_ERR_LAUNCH;
    ws = side_stream;
  }
  if (r > 1) {                                   // the reduction branch's two problems over its B * Lk tokens
    const void* dred = srn ? a->dred : a->dkvin;
    const int K = r * r * C;
    const void* dys[2] = {a->dkv, dred};
    const void* xs[2] = {a->kvin, a->patches};
    float* dWs[2] = {a->dWkv, a->dWsr};
    float* dbs[2] = {nullptr, a->dbsr};
    const int Ns[2] = {2 * C, C}, Ks[2] = {C, K};
    const int64_t ldy[2] = {2 * C, C}, ldx[2] = {C, K};
    const float* rs[2] = {nullptr, nullptr};
    rc = TCALL(VTX_T_WGRAD_SR_B, rows, C, K, 0, ws,
               vtx_wgrad_group(dt, 2, dys, xs, dWs, dbs, Ns, Ks, ldy, ldx, rs, 1, 0.f, rows, a->wgrad2_ws, a->wgrad2_ws_bytes, 0, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 0, ws));
    if (rc) return rc;
  }
  const int n = r > 1 ? 4 : 5;
  const void* dys[5] = {a->dy, a->dz, a->dx1, a->dq, a->dkv};
  const void* xs[5] = {a->h, a->ln2, a->o, a->ln1, a->ln1};
  float* dWs[5] = {a->dW2, a->dW1, a->dWo, a->dWq, a->dWkv};
  float* dbs[5] = {a->db2, a->db1, a->dbo, nullptr, nullptr};
  const int Ns[5] = {C, ff, C, C, 2 * C}, Ks[5] = {ff, C, C, C, C};
  const int64_t ldy[5] = {C, ff, C, C, 2 * C}, ldx[5] = {ff, C, C, C, C};
  const float* rs[5] = {a->s2, nullptr, a->s1, nullptr, nullptr};
  const int ncol = srn ? 3 : 2;
  const float* cpart[3] = {(const float*)a->ln2_ws, (const float*)a->ln1_ws, (const float*)a->lns_ws};
  float* cout0[3] = {a->dg2, a->dg1, a->dgs};
  float* cout1[3] = {a->dbe2, a->dbe1, a->dbs};
  const int cnb[3] = {vtx_layernorm_bwd_blocks(M, C), vtx_layernorm_bwd_blocks(M, C), srn ? vtx_layernorm_bwd_blocks(rows, C) : 0};
  const int cC[3] = {C, C, C};
  const int cld[3] = {2 * C, 2 * C, 2 * C};
  return TCALL(VTX_T_WGRAD_SR_A, M, C, ff, r > 1 ? 0 : 1, ws,
               vtx_wgrad_group(dt, n, dys, xs, dWs, dbs, Ns, Ks, ldy, ldx, rs, rps, a->scale_const, M, a->wgrad_ws, a->wgrad_ws_bytes, ncol,
                               cpart, cout0, cout1, cnb, cC, cld, 0, ws));
}

}  // extern "C"
