// Launch plans: a recorded sequence of vfm_* calls replayed on one stream by ONE call from the host language (include/vfmseg_hip.h,
// "launch plans").  Host code only: every entry goes through the same extern "C" entry point, with the same arguments, as when the
// Python binding issues it - the plan removes the per-launch cost of the binding (descriptor filling, ctypes marshalling, tensor
// allocation: ~14 us per launch, 12 ms per train step) and nothing else.
#include <vector>

#include "common.h"

namespace {
struct ProfRec {
  int kind;
  double flops;
  hipEvent_t e0, e1;
};
int g_prof_every = 0;
long g_prof_count = 0;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;   // events are recycled: creating one costs ~5 us

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

int run_one(const vfm_plan_op& o, uint64_t seed, uint64_t rng_base, void* s) {
  switch (o.kind) {
    case VFM_OP_GEMM: return vfm_gemm(&o.u.gemm, s);
    case VFM_OP_ATTN_FWD: return vfm_attn_fwd(&o.u.attn, s);
    case VFM_OP_ATTN_BWD: return vfm_attn_bwd(&o.u.attn, s);
    case VFM_OP_LN_FWD: {
      const auto& a = o.u.ln_fwd;
      return vfm_layernorm_fwd(a.x, a.ld_x, a.w, a.b, a.eps, a.y, a.y_dt, a.ld_y, a.stats, a.rows, a.C, s);
    }
    case VFM_OP_LN_DROPOUT_FWD: {
      const auto& a = o.u.ln_drop;
      return vfm_layernorm_dropout_fwd(a.x, a.ld_x, a.w, a.b, a.eps, a.y, a.ld_y, a.stats, a.y_drop, a.ld_yd, a.mask, a.ld_mask, a.p, seed,
                                       rng_base + a.offset, a.rows, a.C, s);
    }
    case VFM_OP_LN_BWD_SCALED: {
      const auto& a = o.u.ln_bwd;
      return vfm_layernorm_bwd_scaled(a.dy, a.dy_dt, a.ld_dy, a.x, a.ld_x, a.w, a.stats, a.dx, a.ld_dx, a.accumulate_dx, a.t_out, a.ld_t,
                                      a.t_scale, a.rows, a.C, s);
    }
    case VFM_OP_CAST: {
      const auto& a = o.u.cast;
      return vfm_cast(a.src, a.src_dt, a.ld_src, a.dst, a.dst_dt, a.ld_dst, a.rows, a.cols, a.colscale, s);
    }
    case VFM_OP_STRIDED_COPY: {
      const auto& a = o.u.copy;
      return vfm_strided_copy(a.src, a.src_dt, a.dst, a.dst_dt, a.n[0], a.n[1], a.n[2], a.n[3], a.ss[0], a.ss[1], a.ss[2], a.ss[3], a.ds[0],
                              a.ds[1], a.ds[2], a.ds[3], a.accumulate, s);
    }
    default: VFM_FAIL(VFM_E_INVAL, "vfm_run_plan: unknown op kind %d", o.kind);
  }
}
}  // namespace

extern "C" int vfm_run_plan(const vfm_plan_op* ops, int n, uint64_t seed, uint64_t rng_base, void* stream) {
  VFM_CHECK(ops != nullptr || n == 0, VFM_E_INVAL, "vfm_run_plan: null plan");
  hipStream_t hs = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    const vfm_plan_op& o = ops[i];
    bool timed = false;
    ProfRec rec;
    if (g_prof_every > 0 && o.prof_kind != VFM_PROF_NONE && (++g_prof_count % g_prof_every) == 0 && g_prof_recs.size() < 65536) {
      rec.kind = o.prof_kind, rec.flops = o.flops, rec.e0 = prof_event(), rec.e1 = prof_event();
      timed = rec.e0 && rec.e1 && hipEventRecord(rec.e0, hs) == hipSuccess;
    }
    const int rc = run_one(o, seed, rng_base, stream);
    if (timed) {
      if (hipEventRecord(rec.e1, hs) == hipSuccess) g_prof_recs.push_back(rec);
    }
    if (rc != VFM_OK) {
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", g_vfm_err);
      VFM_FAIL(rc, "vfm_run_plan: entry %d (kind %d) failed: %s", i, o.kind, msg);
    }
  }
  return VFM_OK;
}

extern "C" int vfm_prof_config(int every) {
  VFM_CHECK(every >= 0, VFM_E_INVAL, "vfm_prof_config: every >= 0");
  g_prof_every = every;
  g_prof_count = 0;
  return VFM_OK;
}

extern "C" int vfm_prof_read(double* out, int cap) {
  VFM_CHECK(out != nullptr || cap == 0, VFM_E_INVAL, "vfm_prof_read: null buffer");
  int n = 0;
  for (const ProfRec& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && n < cap) {
      out[3 * n] = (double)r.kind, out[3 * n + 1] = r.flops, out[3 * n + 2] = (double)ms;
      ++n;
    }
    g_prof_pool.push_back(r.e0), g_prof_pool.push_back(r.e1);
  }
  g_prof_recs.clear();
  return n;
}
