// C ABI (include/camo_fusion.h): workspace carving and the launch schedules of the fused
// forward / backward of the fusion model.  No device code here.
//
// Algebra used (results equal to the reference up to fp32 re-association):
//   * mean-pool linearity.  The reference computes Z = Y + FFN(Y) for every node and then only
//     uses mean_t Z (fusion_model.py:120,134).  Since the second FFN layer is linear,
//       mean_t Z = mean_t Y + (mean_t H1d) . W2^T + b2,   H1d = dropout(relu(Y.W1^T + b1)),
//     so the [Nr,512]x[512,256] GEMM per sample (22 % of the forward FLOPs) becomes a
//     [B,512]x[512,256] one, and in the backward d(H1d) is one row per sample broadcast
//     through the ReLU/dropout mask, dW2 = d(pool)^T . mean(H1d)  (two more full-size GEMMs gone).
//   * K|V projections share their input, so they run as one N=2H GEMM on the contiguous rows
//     H..3H of in_proj_weight; their input gradients as one K=2H GEMM.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/camo_fusion.h"

#include "attn.h"
#include "gemm.h"
#include "fused_rows.h"
#include "tail_wide.h"
#include "gemm16.h"
#include "misc.h"
#include "rg_gnn.h"
#include "rg_features.h"
#include "../../include/camo_rg_gnn.h"
#include "../../include/camo_rg_features.h"

namespace {

thread_local std::string g_err;
// Schedule options: the CALLER's (camo_options_t behind camo_dims_t::options), bound for the duration of one entry-point call.  No
// option state lives in the library: two engines in one process cannot change each other's schedule.
constexpr camo_options_t k_default_options = {/*sched16*/ -1, /*fused*/ -1, /*tail17*/ -1, /*fused_rt*/ -1, /*wide2*/ -1, /*fused_one*/ 1, /*wide_front_rt*/ 0,
                                              /*tailw*/ -1, /*tailw_bwd*/ -1, /*param_space*/ -1, /*tn_big*/ -1, /*fused_variant*/ 1, /*back_lead*/ 1,
                                              /*tn_balance*/ 1, /*tn_kcap*/ 0, /*tn_exp*/ 0, /*exp*/ 0, /*fused_save*/ 0, /*tail_skip_arrival*/ 0, /*wide2_bwd*/ -1};
static thread_local const camo_options_t* t_opt = &k_default_options;
struct OptScope {                     // binds the caller's options (and the other translation units' per-call copies) for one entry-point call
  const camo_options_t* prev;
  explicit OptScope(const camo_dims_t* d) : prev(t_opt) {
    t_opt = (d && d->options) ? d->options : &k_default_options;
    g_back_lead_mode = t_opt->back_lead; g_gemm16_balance = t_opt->tn_balance; g_gemm16_tn_kcap = t_opt->tn_kcap; g_gemm16_exp = t_opt->tn_exp;
    g_gemm16_tn_big = t_opt->tn_big;
    if (d && d->options && d->options->tail_skip_arrival) { g_tail_debug_skip = d->options->tail_skip_arrival; d->options->tail_skip_arrival = 0; }
  }
  ~OptScope() { t_opt = prev; }
};
#define g_opt_sched16 (t_opt->sched16)
#define g_opt_fused (t_opt->fused)
#define g_opt_param_space (t_opt->param_space)
#define g_opt_tail17 (t_opt->tail17)
#define g_opt_fused_variant (t_opt->fused_variant)
#define g_opt_fused_rt (t_opt->fused_rt)
#define g_opt_wide2 (t_opt->wide2)
#define g_opt_wide2_bwd (t_opt->wide2_bwd)
#define g_opt_fused_one (t_opt->fused_one)
#define g_opt_wide_front_rt (t_opt->wide_front_rt)
#define g_opt_tailw_bwd (t_opt->tailw_bwd)
#define g_opt_tailw (t_opt->tailw)
#define g_opt_exp (t_opt->exp)
#define g_opt_fused_save (t_opt->fused_save)
unsigned long long* g_dbg_stamps = nullptr;   // developer timeline buffer of the fused kernels (camo_debug_set_stamps; like camo_prof_*: a profiling facility, not a schedule option)
int g_dbg_stamp_blocks = 0;
static thread_local bool t_tailw_bwd_planes = false;   // set by a training forward that built the tail's transposed planes (this call's workspace)

int fail(int code, const std::string& msg) { g_err = msg; return code; }
int fail_hip(int e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString((hipError_t)e);
  return CAMO_E_HIP;
}
#define CK(x, where)                         \
  do {                                       \
    int e_ = (x);                            \
    if (e_ != 0) return fail_hip(e_, where); \
  } while (0)

typedef unsigned short us;
struct Carver {
  char* base; size_t off;
  explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// The batch descriptor (camo_prepare_batch): [ row -> sample map | 1 / Nr | first 32-row tile of every sample ]
struct Desc { int* row_sample; float* inv_nr; int* tile_off; int4* tile_desc; size_t bytes; };
Desc desc_carve(int B, int T, void* base) {
  Desc d{};
  Carver c(base);
  d.row_sample = c.take<int>((size_t)T); d.inv_nr = c.take<float>((size_t)B); d.tile_off = c.take<int>((size_t)B + 1);
  d.tile_desc = c.take<int4>((size_t)T / 32 + B);       // one entry per block of the fused kernels' RG tile range
  c.off = (c.off + 255) & ~size_t(255);
  d.bytes = c.off;
  return d;
}

// arrival-counter words of the one-launch tail behind its all-reduce buffers: 4 per group of 16 samples (misc.hip, tail_fused_kernel)
static inline size_t tail_counter_words(int B) { return (size_t)4 * ((B + 15) / 16 > 0 ? (B + 15) / 16 : 1); }

struct Ws {
  // zeroed once per forward: [ means | dfused | dKV ] (accumulated into by atomics)
  float* zero_base; size_t zero_bytes;
  // forward (saved for backward)
  float *R, *G, *Q, *KV2, *KV, *Q2, *P, *P2, *O, *O2, *U, *U2, *st1, *st2, *Y, *Y2, *H1, *H2;
  float *means, *Ymean, *H1mean, *Y2mean, *H2mean; size_t means_n;
  float *comb, *F1, *fused, *hid, *a2;
  // backward scratch
  float *dlog, *dhid, *dfused, *dF1, *dcomb, *dHm1, *dHm2, *da2;
  float *dH1, *dH2, *dY, *dY2, *dU, *dU2, *dO, *dO2, *dQ, *dKV, *dQ2, *dKV2, *dS2, *dR, *dG;
  // bf16 schedule ("sched16"): a bf16 copy of every node-level GEMM operand.  Activations have their row
  // count padded to a multiple of 128 (Tp, TKp; the pad rows are cleared by the prep launch) so the weight-
  // gradient GEMMs contract over whole 64-row tiles without masks.  Weights: [out][in] copies for x.W^T,
  // transposed copies for dy.W, and the two in-projection slices that meet at one input concatenated
  // (WcRgT = [Wq1^T | Wk2^T | Wv2^T], WcKgT = [Wq2^T | Wk1^T | Wv1^T], both [H][3H]).
  struct H16 {
    unsigned short *X, *KG, *R, *G, *O, *O2, *Y, *Y2, *dH1, *dH2, *dU, *dU2, *dQKV, *dQKVkg, *dR, *dG;
    unsigned short *H1, *H2;   // post-ReLU/dropout FFN activations: only their sign pattern is read again (backward mask)
    unsigned short *Wrg, *Wkg, *Win1, *Win2, *Wo1, *Wo2, *W1, *W2, *W1T, *W2T, *Wo1T, *Wo2T, *WcRgT, *WcKgT;
  } h;
  size_t Tp, TKp;
  // fused row-tile schedule (fused_rows.h): weight shadows in MFMA-fragment order and the bf16 activations that cross
  // its launches / are saved for backward.  Only carved at the reference configuration (fused17_dims).
  struct F17 {
    us16 *Wrg, *Wkg, *Wqkv_rg, *Wqkv_kg, *Wo1, *Wo2, *W1, *W2;
    us16* Wf_rg; float* bf_rg;          // the RG rows' folded in-projection (fused_wide2.hip, launch_fold_rg): shadow of [768 x 128], bias [768]
    us16 *X16, *KG16, *R16, *G16, *Q16, *Q2_16, *KV16, *KV2_16, *O16, *O2_16, *Y16, *Y2_16, *XH16, *XH2_16;
    float *rstd1, *rstd2, *lse2, *part; uint32_t *mask1, *mask2;
    // backward: transposed shadows, the bf16 gradients that are weight-gradient operands, per-sample exchange buffers
    us16 *W1T, *W2T, *Wo1T, *Wo2T, *WcRgT, *WcKgT;
    us16 *dH16, *dH2_16, *dU16, *dU2_16, *dQKV16, *dQKVkg16, *dR16, *dG16, *dO2_16;
    float *delta2, *dGpart;
    us16* tailw[20];      // hi / lo planes of the per-sample tail's weights (tail_wide.h): W13, W23, Wfu0 [256 x 512], Wfu3 [256 x 256], heads [512 x 256];
                          // 10..19: of their transposes for the tail's backward: heads^T [256 x 512], Wfu3^T [256 x 256], Wfu0^T, W13^T, W23^T [512 x 256]
  } f;
  int* tickets;         // [2][B] arrival counters (forward: KG->RG attention splits; backward: a sample's RG tiles), in the zero block
  float* dQ2acc;        // [TK][H] fp32 sums of the KG->RG query gradient (in the zero block, fused backward)
  float* parM;          // [2][3H][D] + [2][3H]: dQKV^T x and colsum(dQKV) of the two streams (in the zero block)
  float* tailsum;       // [B][4H] all-reduce buffers of the one-launch tail (F1 | hidden | dF1) + 4 counter words per 16 samples (in the zero block)
  size_t bytes;
};

// the fused row-tile kernels are written for the reference configuration
bool fused17_dims(const camo_dims_t& d) {
  return d.fusion_type == CAMO_FUSION_CROSS_ATTENTION && d.hidden_dim == 256 && d.num_heads == 8 && d.rg_dim == 128 && d.kg_dim == 128;
}

Ws carve(const camo_dims_t& d, int B, int T, int Nk, void* base) {
  Ws w{};
  Carver c(base);
  const size_t H = d.hidden_dim, TK = (size_t)B * Nk, nh = d.num_heads, Wd = 2 * d.num_classes + 2;
  if (d.fusion_type == CAMO_FUSION_CROSS_ATTENTION) {
    const size_t Fh = H / 2;
    w.R = c.take<float>(T * H); w.G = c.take<float>(TK * H);
    w.Q = c.take<float>(T * H); w.KV2 = c.take<float>(T * 2 * H);
    w.KV = c.take<float>(TK * 2 * H); w.Q2 = c.take<float>(TK * H);
    w.P = c.take<float>(T * nh * Nk); w.P2 = c.take<float>(T * nh * Nk);
    w.O = c.take<float>(T * H); w.O2 = c.take<float>(TK * H);
    w.U = c.take<float>(T * H); w.U2 = c.take<float>(TK * H);
    w.st1 = c.take<float>(T * 2); w.st2 = c.take<float>(TK * 2);
    w.Y = c.take<float>(T * H); w.Y2 = c.take<float>(TK * H);
    w.H1 = c.take<float>(T * 2 * H); w.H2 = c.take<float>(TK * 2 * H);
    w.means_n = (size_t)B * 6 * H;
    {   // one contiguous block so a single memset clears every atomically-accumulated buffer
      const size_t nzt = ((size_t)2 * B + 3) & ~size_t(3);                   // tickets: padded to 16 bytes
      const size_t npar = (size_t)2 * (3 * H * d.rg_dim + 3 * H) + 8;        // per stream: dQKV^T x [3H][D] and colsum(dQKV) [3H] (fused backward, parameter space)
      const size_t nz = w.means_n + (size_t)B * H + nzt + TK * 2 * H + TK * H + (size_t)B * 4 * H + ((tail_counter_words(B) + 3) & ~size_t(3)) + npar;
      float* z = c.take<float>(nz);
      w.zero_base = z; w.zero_bytes = nz * sizeof(float);
      w.means = z; w.dfused = z ? z + w.means_n : nullptr;
      w.tickets = z ? reinterpret_cast<int*>(w.dfused + (size_t)B * H) : nullptr;
      w.dKV = z ? w.dfused + (size_t)B * H + nzt : nullptr;
      w.dQ2acc = z ? w.dKV + TK * 2 * H : nullptr;
      w.tailsum = z ? w.dQ2acc + TK * H : nullptr;
      w.parM = z ? w.tailsum + (size_t)B * 4 * H + ((tail_counter_words(B) + 3) & ~size_t(3)) : nullptr;
    }
    if (w.means) { w.Ymean = w.means; w.H1mean = w.Ymean + B * H; w.Y2mean = w.H1mean + B * 2 * H; w.H2mean = w.Y2mean + B * H; }
    w.comb = c.take<float>(B * 2 * H); w.F1 = c.take<float>(B * H); w.fused = c.take<float>(B * H);
    w.hid = c.take<float>(B * 4 * Fh);
    w.dlog = c.take<float>(B * Wd); w.dhid = c.take<float>(B * 4 * Fh);
    w.dF1 = c.take<float>(B * H); w.dcomb = c.take<float>(B * 2 * H);
    w.dHm1 = c.take<float>(B * 2 * H); w.dHm2 = c.take<float>(B * 2 * H);
    w.dH1 = c.take<float>(T * 2 * H); w.dH2 = c.take<float>(TK * 2 * H);
    w.dY = c.take<float>(T * H); w.dY2 = c.take<float>(TK * H);
    w.dU = c.take<float>(T * H); w.dU2 = c.take<float>(TK * H);
    w.dO = c.take<float>(T * H); w.dO2 = c.take<float>(TK * H);
    w.dQ = c.take<float>(T * H);
    w.dQ2 = c.take<float>(TK * H); w.dKV2 = c.take<float>(T * 2 * H);
    w.dS2 = c.take<float>(T * nh * Nk);
    w.dR = c.take<float>(T * H); w.dG = c.take<float>(TK * H);
    {
      const size_t Tp = ((size_t)T + 127) / 128 * 128, TKp = (TK + 127) / 128 * 128, D = d.rg_dim, Dk = d.kg_dim;
      w.Tp = Tp; w.TKp = TKp;
      Ws::H16& h = w.h;
      h.X = c.take<us>(Tp * D); h.KG = c.take<us>(TKp * Dk); h.R = c.take<us>(Tp * H); h.G = c.take<us>(TKp * H);
      h.O = c.take<us>(Tp * H); h.O2 = c.take<us>(TKp * H); h.Y = c.take<us>(Tp * H); h.Y2 = c.take<us>(TKp * H);
      h.dH1 = c.take<us>(Tp * 2 * H); h.dH2 = c.take<us>(TKp * 2 * H); h.dU = c.take<us>(Tp * H); h.dU2 = c.take<us>(TKp * H);
      h.dQKV = c.take<us>(Tp * 3 * H); h.dQKVkg = c.take<us>(TKp * 3 * H); h.dR = c.take<us>(Tp * H); h.dG = c.take<us>(TKp * H);
      h.Wrg = c.take<us>(H * D); h.Wkg = c.take<us>(H * Dk); h.Win1 = c.take<us>(3 * H * H); h.Win2 = c.take<us>(3 * H * H);
      h.Wo1 = c.take<us>(H * H); h.Wo2 = c.take<us>(H * H); h.W1 = c.take<us>(2 * H * H); h.W2 = c.take<us>(2 * H * H);
      h.W1T = c.take<us>(2 * H * H); h.W2T = c.take<us>(2 * H * H); h.Wo1T = c.take<us>(H * H); h.Wo2T = c.take<us>(H * H);
      h.WcRgT = c.take<us>(3 * H * H); h.WcKgT = c.take<us>(3 * H * H);
      h.H1 = c.take<us>(Tp * 2 * H); h.H2 = c.take<us>(TKp * 2 * H);      // (padded: they double as weight-gradient operands)
      if (fused17_dims(d)) {
        Ws::F17& f = w.f;
        f.Wrg = c.take<us>(H * D); f.Wkg = c.take<us>(H * Dk); f.Wqkv_rg = c.take<us>(3 * H * H); f.Wqkv_kg = c.take<us>(3 * H * H);
        f.Wo1 = c.take<us>(H * H); f.Wo2 = c.take<us>(H * H); f.W1 = c.take<us>(2 * H * H); f.W2 = c.take<us>(2 * H * H);
        f.X16 = c.take<us>(Tp * D); f.KG16 = c.take<us>(TKp * Dk); f.R16 = c.take<us>(Tp * H); f.G16 = c.take<us>(TKp * H);
        f.Q16 = c.take<us>(Tp * H); f.Q2_16 = c.take<us>(TKp * H); f.KV16 = c.take<us>(TKp * 2 * H); f.KV2_16 = c.take<us>(Tp * 2 * H);
        f.O16 = c.take<us>(Tp * H); f.O2_16 = c.take<us>(TKp * H); f.Y16 = c.take<us>(Tp * H); f.Y2_16 = c.take<us>(TKp * H);
        f.XH16 = c.take<us>(Tp * H); f.XH2_16 = c.take<us>(TKp * H);
        f.rstd1 = c.take<float>(Tp); f.rstd2 = c.take<float>(TKp); f.lse2 = c.take<float>((size_t)B * 8 * 16 * 2);
        f.mask1 = c.take<uint32_t>(Tp * 16); f.mask2 = c.take<uint32_t>(TKp * 16);
        f.part = c.take<float>(((size_t)T / 32 + B + 2) * 8 * FUSED_PART_FLOATS);
        f.W1T = c.take<us>(2 * H * H); f.W2T = c.take<us>(2 * H * H); f.Wo1T = c.take<us>(H * H); f.Wo2T = c.take<us>(H * H);
        f.WcRgT = c.take<us>(3 * H * H); f.WcKgT = c.take<us>(3 * H * H);
        f.dH16 = c.take<us>(Tp * 2 * H); f.dH2_16 = c.take<us>(TKp * 2 * H); f.dU16 = c.take<us>(Tp * H); f.dU2_16 = c.take<us>(TKp * H);
        f.dQKV16 = c.take<us>(Tp * 3 * H); f.dQKVkg16 = c.take<us>(TKp * 3 * H); f.dR16 = c.take<us>(Tp * H); f.dG16 = c.take<us>(TKp * H);
        f.dO2_16 = c.take<us>(TKp * H); f.delta2 = c.take<float>((size_t)B * 8 * 16); f.dGpart = c.take<float>(TKp * H);
        for (int i = 0; i < 10; ++i) f.tailw[i] = c.take<us>(i < 6 ? 2 * H * H : (i < 8 ? H * H : 2 * H * H));
        for (int i = 10; i < 20; ++i) f.tailw[i] = c.take<us>((i == 12 || i == 13) ? H * H : 2 * H * H);
        f.Wf_rg = c.take<us>(3 * H * D); f.bf_rg = c.take<float>(3 * H);
      }
    }
  } else {
    const size_t F = H / 2, Fh = F / 2, Dc = (size_t)d.rg_dim + d.kg_dim;
    w.means_n = (size_t)B * Dc;
    {
      const size_t nz = w.means_n + (size_t)B * F;
      float* z = c.take<float>(nz);
      w.zero_base = z; w.zero_bytes = nz * sizeof(float);
      w.comb = z;                                // [B, rg_dim+kg_dim] = the two means, zeroed then accumulated
      w.dfused = z ? z + w.means_n : nullptr;
    }
    w.means = w.comb;
    w.F1 = c.take<float>(B * H);               // a1
    w.a2 = c.take<float>(B * F);
    w.fused = c.take<float>(B * F);
    w.hid = c.take<float>(B * 4 * Fh);
    w.dlog = c.take<float>(B * Wd); w.dhid = c.take<float>(B * 4 * Fh);
    w.da2 = c.take<float>(B * F); w.dF1 = c.take<float>(B * H);
  }
  c.off = (c.off + 255) & ~size_t(255);
  w.bytes = c.off;
  return w;
}

int check_dims(const camo_dims_t* d, int B, int T, int Nk) {
  if (!d) return fail(CAMO_E_ARG, "dims is null");
  if (B < 1 || T < B || Nk < 1) return fail(CAMO_E_ARG, "need B >= 1, T >= B (every sample has >= 1 RG row), Nk >= 1");
  if (d->rg_dim < 1 || d->kg_dim < 1 || d->hidden_dim < 4 || d->num_classes < 1 || d->num_classes > 64)
    return fail(CAMO_E_ARG, "bad model dimensions");
  if (!(d->dropout >= 0.f && d->dropout < 1.f)) return fail(CAMO_E_ARG, "dropout must be in [0,1)");
  if (d->fusion_type == CAMO_FUSION_CROSS_ATTENTION) {
    if (d->num_heads < 1 || d->hidden_dim % d->num_heads) return fail(CAMO_E_ARG, "hidden_dim must be divisible by num_heads");
    if (d->hidden_dim % 2) return fail(CAMO_E_UNSUPPORTED, "hidden_dim must be even");
    if (!attn_supported(d->hidden_dim, d->num_heads, Nk))
      return fail(CAMO_E_UNSUPPORTED, "attention kernels support num_heads <= 256, head_dim <= 256, Nk <= 64 within 160 KB of LDS");
    if (!ln_supported(d->hidden_dim)) return fail(CAMO_E_UNSUPPORTED, "LayerNorm kernels support hidden_dim <= 1024");
  } else if (d->fusion_type == CAMO_FUSION_LATE) {
    if (d->hidden_dim % 4) return fail(CAMO_E_UNSUPPORTED, "late fusion needs hidden_dim divisible by 4");
  } else {
    return fail(CAMO_E_ARG, "unknown fusion_type");
  }
  if ((double)T * d->hidden_dim * 2 > 2.0e9 || (double)T * d->num_heads * Nk > 4.0e9)
    return fail(CAMO_E_UNSUPPORTED, "batch too large for 32-bit element indices");
  return 0;
}

struct GB {
  GemmBatch b;
  int prec; hipStream_t st;
  GB(const DropCfg& d, int prec_, hipStream_t st_) : prec(prec_), st(st_) { std::memset(&b, 0, sizeof(b)); b.drop = d; }
  GemmProb& add(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K, int flags) {
    GemmProb& p = b.p[b.n++];
    std::memset(&p, 0, sizeof(p));   // slots are reused across launches: no stale res/bias_grad/flags
    p.A = A; p.lda = lda; p.B = Bm; p.ldb = ldb; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.flags = flags;
    p.aux_scale = 1.f;
    return p;
  }
  // y = x.W^T (+bias): x [M,K], W [N,K]
  GemmProb& nt(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, int ldy, int M, int N, int K, int flags = 0) {
    GemmProb& p = add(x, ldx, W, ldw, y, ldy, M, N, K, flags);
    p.bias = bias;
    return p;
  }
  // dx = dy.W : dy [M,K=Nout], W [Nout, N=in]
  GemmProb& nn(const float* dy, int lddy, const float* W, int ldw, float* dx, int lddx, int M, int N, int K, int flags = 0) {
    return add(dy, lddy, W, ldw, dx, lddx, M, N, K, flags | GF_B_KMAJOR);
  }
  // dW [Nout, Nin] += dy^T.x : dy [rows, Nout], x [rows, Nin]; db [Nout] += colsum(dy)
  GemmProb& tn(const float* dy, int lddy, const float* x, int ldx, float* dW, int lddw, float* db, int Nout, int Nin, int rows) {
    GemmProb& p = add(dy, lddy, x, ldx, dW, lddw, Nout, Nin, rows, GF_A_KMAJOR | GF_B_KMAJOR | GF_ATOMIC);
    p.bias_grad = db;
    return p;
  }
  int run() {
    if (b.n == 0) return 0;
    int e = launch_gemm_batch(b, prec, st);
    b.n = 0;
    return e;
  }
};
void set_res(GemmProb& p, const float* res, int ldr) { p.res = res; p.ldr = ldr; }
void set_drop(GemmProb& p, uint32_t site) { p.flags |= GF_DROPOUT; p.drop_site = site; }
void set_relu_bwd(GemmProb& p, const float* act, int ldr, float scale) {
  p.flags |= GF_RELU_BWD; p.res = act; p.ldr = ldr; p.aux_scale = scale;
}
void set_bcast(GemmProb& p, const float* v, int ldv, const int* row_sample, const float* inv_nr, int uniform_n) {
  p.flags |= GF_RES_BCAST; p.res = v; p.ldr = ldv; p.row_sample = row_sample; p.inv_nr = inv_nr; p.uniform_n = uniform_n;
}

// ---- the bf16 schedule's GEMM batch (gemm16.h) ------------------------------------------------
struct GB16 {
  Gemm16Batch b; hipStream_t st;
  GB16(const DropCfg& d, hipStream_t st_) : st(st_) { std::memset(&b, 0, sizeof(b)); b.drop = d; }
  Gemm16Prob& add() { Gemm16Prob& p = b.p[b.n++]; std::memset(&p, 0, sizeof(p)); p.aux_scale = 1.f; return p; }
  // y = x.W^T (+bias): x16 [M,K], W16 [N,K]; fp32 result y and/or bf16 result y16 (either may be null)
  Gemm16Prob& nt(const us* x, int ldx, const us* W, int ldw, const float* bias, float* y, int ldy, us* y16, int ldy16,
                 int M, int N, int K, int flags = 0) {
    Gemm16Prob& p = add();
    p.A = x; p.lda = ldx; p.B = W; p.ldb = ldw; p.bias = bias; p.C = y; p.ldc = ldy; p.C16 = y16; p.ldc16 = ldy16;
    p.M = M; p.N = N; p.K = K; p.flags = flags;
    return p;
  }
  // dW [Nout, Nin] += dy^T.x : dy16 [rows, Nout], x16 [rows, Nin]; db [Nout] += colsum(dy)
  Gemm16Prob& tn(const us* dy, int lddy, const us* x, int ldx, float* dW, int lddw, float* db, int Nout, int Nin, int rows) {
    Gemm16Prob& p = add();
    p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx; p.C = dW; p.ldc = lddw; p.bias_grad = db;
    p.M = Nout; p.N = Nin; p.K = rows; p.flags = GF_A_KMAJOR | GF_B_KMAJOR | GF_ATOMIC;
    return p;
  }
  int run() {
    if (b.n == 0) return 0;
    int e = launch_gemm16_batch(b, st);
    b.n = 0;
    return e;
  }
};
void set_res(Gemm16Prob& p, const float* res, int ldr) { p.res = res; p.ldr = ldr; }
void set_drop(Gemm16Prob& p, uint32_t site) { p.flags |= GF_DROPOUT; p.drop_site = site; }
void set_bcast(Gemm16Prob& p, const float* v, int ldv, const int* row_sample, const float* inv_nr, int uniform_n) {
  p.flags |= GF_RES_BCAST; p.res = v; p.ldr = ldv; p.row_sample = row_sample; p.inv_nr = inv_nr; p.uniform_n = uniform_n;
}

// The bf16 schedule runs when the operands can live in HBM as bf16 tiles the gemm16 kernel takes whole:
// cross-attention fusion with both input projections, every width a multiple of 64, head_dim 32 attention
// on the MFMA kernels.  Anything else (and CAMO_SCHED16=0) takes the general fp32-operand schedule.
bool sched16_ok(const camo_dims_t& d, const float* const* P, int precision, int T, int Nk, int max_nr) {
  if (g_opt_sched16 == 0 || precision != CAMO_PREC_BF16 || d.fusion_type != CAMO_FUSION_CROSS_ATTENTION) return false;
  if (!P[CAMO_P_RG_PROJ_W] || !P[CAMO_P_KG_PROJ_W]) return false;
  if ((d.hidden_dim % 64) || (d.rg_dim % 64) || (d.kg_dim % 64)) return false;
  if (!attn_mfma_ok(d.hidden_dim, d.num_heads, Nk, max_nr, false) || !attn_mfma_ok(d.hidden_dim, d.num_heads, Nk, max_nr, true))
    return false;
  return ((double)T + 128.0) * 3.0 * d.hidden_dim * 2.0 < 4.0e9;
}

// ---- the four heads (fusion_model.py:208-235), shared by both fusion types -------------------
// The per-sample ("tail") GEMMs have M = B rows and a negligible FLOP share, so they always run
// on the exact f32 MFMA; `precision` selects the MFMA type of the node-level (T-row) GEMMs only.
// labels + outputs of the native training call: when given, the head output layer, the loss and the head output
// layer's backward run as one kernel (misc.hip, heads_loss_kernel) instead of three launches
struct FusedLoss { const int64_t* y; const float* e; const float* s; float* terms; int32_t* pred; float* const* head_grads; };

int heads_forward(const camo_dims_t& d, const float* const* hp /*16 pointers*/, const Ws& w, int B, int F,
                  float* outs, GB& g, const FusedLoss* fl = nullptr, bool hidden_done = false, bool defer_out_grads = false) {
  const int Fh = F / 2, C = d.num_classes, Wd = 2 * C + 2;
  const int nout[4] = {C, C, 1, 1}, coff[4] = {0, C, 2 * C, 2 * C + 1};
  if (!hidden_done) {                      // (else w.hid came out of the two-plane tail launch)
    for (int x = 0; x < 4; ++x) {
      GemmProb& p = g.nt(w.fused, F, hp[4 * x], F, hp[4 * x + 1], w.hid + x * Fh, 4 * Fh, B, Fh, F, GF_RELU);
      set_drop(p, SITE_HEAD0 + x);
    }
    CK(g.run(), "heads hidden");
  }
  if (fl) {
    HeadsOut ho;
    for (int x = 0; x < 4; ++x) { ho.W[x] = hp[4 * x + 2]; ho.b[x] = hp[4 * x + 3]; ho.gW[x] = fl->head_grads[4 * x + 2]; ho.gb[x] = fl->head_grads[4 * x + 3]; }
    CK(launch_heads_loss(w.hid, ho, reinterpret_cast<const long long*>(fl->y), fl->e, fl->s, B, C, Fh, g.b.drop.scale, outs, fl->terms,
                         fl->pred, w.dhid, g.st, defer_out_grads ? w.dlog : nullptr), "heads out + loss + heads out bwd");
    return 0;
  }
  for (int x = 0; x < 4; ++x)
    g.nt(w.hid + x * Fh, 4 * Fh, hp[4 * x + 2], Fh, hp[4 * x + 3], outs + coff[x], Wd, B, nout[x], Fh, x == 3 ? GF_SIGMOID : 0);
  CK(g.run(), "heads out");
  return 0;
}

// d_outs -> dfused (w.dfused, zeroed here) and the 16 head-parameter gradients
int heads_backward(const camo_dims_t& d, const float* const* hp, float* const* hg, const Ws& w, int B, int F,
                   const float* outs, const float* d_outs, int pre_activation, const DropCfg& drop, hipStream_t st, GB& g,
                   bool heads_out_done = false) {
  const int Fh = F / 2, C = d.num_classes, Wd = 2 * C + 2;
  const int nout[4] = {C, C, 1, 1}, coff[4] = {0, C, 2 * C, 2 * C + 1};
  // (w.dfused was zeroed by the forward's memset of the workspace's zero block)
  if (!heads_out_done) {      // else w.dhid and the output-layer gradients came out of heads_loss_kernel
    const float* dlog = d_outs;
    if (!pre_activation) { CK(launch_head_out_grad(outs, d_outs, w.dlog, B, Wd, st), "head_out_grad"); dlog = w.dlog; }
    for (int x = 0; x < 4; ++x) {
      GemmProb& p = g.nn(dlog + coff[x], Wd, hp[4 * x + 2], Fh, w.dhid + x * Fh, 4 * Fh, B, Fh, nout[x]);
      set_relu_bwd(p, w.hid + x * Fh, 4 * Fh, drop.scale);
      g.tn(dlog + coff[x], Wd, w.hid + x * Fh, 4 * Fh, hg[4 * x + 2], Fh, hg[4 * x + 3], nout[x], Fh, B);
    }
    CK(g.run(), "heads out bwd");
  }
  for (int x = 0; x < 4; ++x) {
    g.nn(w.dhid + x * Fh, 4 * Fh, hp[4 * x], F, w.dfused, F, B, F, Fh, GF_ATOMIC);
    g.tn(w.dhid + x * Fh, 4 * Fh, w.fused, F, hg[4 * x], F, hg[4 * x + 1], Fh, F, B);
  }
  CK(g.run(), "heads hidden bwd");
  return 0;
}


// ---- node-level forward of the bf16 schedule: CrossAttentionFusion.forward, fusion_model.py:75-135 ----
int forward_nodes16(const camo_dims_t& d, const float* const* P, const float* rg, const int32_t* rg_offsets,
                    const int32_t* row_sample, const float* inv_nr, const float* kg,
                    int B, int T, int Nk, int max_nr, const Ws& w, float* attn_rg2kg, float* attn_kg2rg, const DropCfg& drop,
                    hipStream_t st) {
  const int H = d.hidden_dim, D = d.rg_dim, Dk = d.kg_dim, TK = B * Nk, nh = d.num_heads;
  const size_t HH = (size_t)H * H;
  const Ws::H16& h = w.h;
  {   // prep: clear the atomics block and the pad rows, cast the inputs and the node-level weights to bf16
    PrepBatch pb; pb.n = 0;
    auto job = [&](int type, const float* src, void* dst, size_t n, int rows, int cols, int ld, int off) {
      PrepJob& J = pb.j[pb.n++];
      J.type = type; J.src = src; J.dst = dst; J.n = n; J.rows = rows; J.cols = cols; J.ld_dst = ld; J.col_off = off; J.blk_begin = 0;
    };
    auto cast = [&](const float* src, us* dst, size_t n) { job(PREP_CAST, src, dst, n, 0, 0, 0, 0); };
    auto castT = [&](const float* src, us* dst, int rows, int cols, int ld, int off) { job(PREP_CAST_T, src, dst, 0, rows, cols, ld, off); };
    auto pad = [&](us* buf, size_t rows, size_t rows_p, size_t width) {
      job(PREP_ZERO, nullptr, buf + rows * width, (rows_p - rows) * width * sizeof(us), 0, 0, 0, 0);
    };
    // the atomics block: pooled means and dfused (this schedule accumulates nothing into dKV)
    job(PREP_ZERO, nullptr, w.zero_base, (size_t)(reinterpret_cast<char*>(w.dKV) - reinterpret_cast<char*>(w.zero_base)), 0, 0, 0, 0);
    cast(rg, h.X, (size_t)T * D); cast(kg, h.KG, (size_t)TK * Dk);
    cast(P[CAMO_P_RG_PROJ_W], h.Wrg, (size_t)H * D); cast(P[CAMO_P_KG_PROJ_W], h.Wkg, (size_t)H * Dk);
    cast(P[CAMO_P_A1_IN_W], h.Win1, 3 * HH); cast(P[CAMO_P_A2_IN_W], h.Win2, 3 * HH);
    cast(P[CAMO_P_A1_OUT_W], h.Wo1, HH); cast(P[CAMO_P_A2_OUT_W], h.Wo2, HH);
    cast(P[CAMO_P_F1_W0], h.W1, 2 * HH); cast(P[CAMO_P_F2_W0], h.W2, 2 * HH);
    castT(P[CAMO_P_F1_W0], h.W1T, 2 * H, H, 2 * H, 0); castT(P[CAMO_P_F2_W0], h.W2T, 2 * H, H, 2 * H, 0);
    castT(P[CAMO_P_A1_OUT_W], h.Wo1T, H, H, H, 0); castT(P[CAMO_P_A2_OUT_W], h.Wo2T, H, H, H, 0);
    castT(P[CAMO_P_A1_IN_W], h.WcRgT, H, H, 3 * H, 0); castT(P[CAMO_P_A2_IN_W] + HH, h.WcRgT, 2 * H, H, 3 * H, H);
    castT(P[CAMO_P_A2_IN_W], h.WcKgT, H, H, 3 * H, 0); castT(P[CAMO_P_A1_IN_W] + HH, h.WcKgT, 2 * H, H, 3 * H, H);
    const size_t t = T, tk = TK;
    pad(h.X, t, w.Tp, D); pad(h.R, t, w.Tp, H); pad(h.O, t, w.Tp, H); pad(h.Y, t, w.Tp, H); pad(h.dH1, t, w.Tp, 2 * H);
    pad(h.dU, t, w.Tp, H); pad(h.dQKV, t, w.Tp, 3 * H); pad(h.dR, t, w.Tp, H); pad(h.H1, t, w.Tp, 2 * H);
    pad(h.KG, tk, w.TKp, Dk); pad(h.G, tk, w.TKp, H); pad(h.O2, tk, w.TKp, H); pad(h.Y2, tk, w.TKp, H); pad(h.dH2, tk, w.TKp, 2 * H);
    pad(h.dU2, tk, w.TKp, H); pad(h.dQKVkg, tk, w.TKp, 3 * H); pad(h.dG, tk, w.TKp, H); pad(h.H2, tk, w.TKp, 2 * H);
    CK(launch_prep(pb, st), "prep (clear + bf16 casts)");
  }
  GB16 g(drop, st);
  // input projections: fp32 for the residual stream, bf16 for the GEMMs that read them
  g.nt(h.KG, Dk, h.Wkg, Dk, P[CAMO_P_KG_PROJ_B], w.G, H, h.G, H, TK, H, Dk);
  g.nt(h.X, D, h.Wrg, D, P[CAMO_P_RG_PROJ_B], w.R, H, h.R, H, T, H, D);
  CK(g.run(), "input projections");
  // in-projections of both attention blocks (packed in_proj_weight: rows 0..H-1 = Wq, H..3H-1 = Wk|Wv)
  g.nt(h.R, H, h.Win1, H, P[CAMO_P_A1_IN_B], w.Q, H, nullptr, 0, T, H, H);
  g.nt(h.R, H, h.Win2 + HH, H, P[CAMO_P_A2_IN_B] + H, w.KV2, 2 * H, nullptr, 0, T, 2 * H, H);
  g.nt(h.G, H, h.Win1 + HH, H, P[CAMO_P_A1_IN_B] + H, w.KV, 2 * H, nullptr, 0, TK, 2 * H, H);
  g.nt(h.G, H, h.Win2, H, P[CAMO_P_A2_IN_B], w.Q2, H, nullptr, 0, TK, H, H);
  CK(g.run(), "attention in-projections");
  // both attention directions in one launch; their outputs are GEMM operands only, so they are written as bf16
  CK(launch_attn_fwd_pair(w.Q, w.KV, w.Q2, w.KV2, rg_offsets, w.P, w.P2, Bf16Dst{h.O, H}, Bf16Dst{h.O2, H}, w.O2, B, max_nr, H, nh, Nk,
                          drop, st), "attention fwd (both directions)");
  if (attn_rg2kg) CK(launch_attn_avg_site(w.P, attn_rg2kg, T, nh, Nk, SITE_ATTN_RG2KG, drop, st), "attn avg rg2kg");
  if (attn_kg2rg) CK(launch_attn_avg(w.P2, attn_kg2rg, T, nh, Nk, drop, st), "attn avg");
  // out-projection + residual (fusion_model.py:119,130), then LayerNorm.  At hidden_dim 256 a block of the GEMM owns
  // whole rows and the LayerNorm (with the mean pool of its output) is its epilogue.
  if (H == 256) {
    Gemm16Prob& p1 = g.nt(h.O, H, h.Wo1, H, P[CAMO_P_A1_OUT_B], w.U, H, h.Y, H, T, H, H);
    set_res(p1, w.R, H);
    p1.ln_mode = 1; p1.ln_gamma = P[CAMO_P_LN1_W]; p1.ln_beta = P[CAMO_P_LN1_B]; p1.ln_stats = w.st1;
    p1.colmean = w.Ymean; p1.ldm = H; p1.row_sample = row_sample; p1.inv_nr = inv_nr;
    Gemm16Prob& p2 = g.nt(h.O2, H, h.Wo2, H, P[CAMO_P_A2_OUT_B], w.U2, H, h.Y2, H, TK, H, H);
    set_res(p2, w.G, H);
    p2.ln_mode = 1; p2.ln_gamma = P[CAMO_P_LN2_W]; p2.ln_beta = P[CAMO_P_LN2_B]; p2.ln_stats = w.st2;
    p2.colmean = w.Y2mean; p2.ldm = H; p2.uniform_n = Nk;
    CK(g.run(), "attention out-projections + layernorm");
  } else {
    set_res(g.nt(h.O, H, h.Wo1, H, P[CAMO_P_A1_OUT_B], w.U, H, nullptr, 0, T, H, H), w.R, H);
    set_res(g.nt(h.O2, H, h.Wo2, H, P[CAMO_P_A2_OUT_B], w.U2, H, nullptr, 0, TK, H, H), w.G, H);
    CK(g.run(), "attention out-projections");
    // (the mean pools of Y and of the FFN activations are accumulated by the kernels that produce them)
    LnSeg s0{w.U, w.Y, w.st1, P[CAMO_P_LN1_W], P[CAMO_P_LN1_B], T, h.Y, w.Ymean, row_sample, inv_nr, 0};
    LnSeg s1{w.U2, w.Y2, w.st2, P[CAMO_P_LN2_W], P[CAMO_P_LN2_B], TK, h.Y2, w.Y2mean, nullptr, nullptr, Nk};
    CK(launch_ln_fwd(s0, s1, H, st), "layernorm fwd");
  }
  // FFN first layers (ReLU + dropout fused), fusion_model.py:53-65.  The activation itself is kept only as bf16 (its
  // sign pattern is the backward mask); its per-sample mean, which the pooled second layer consumes, comes out of
  // the fp32 accumulators in the epilogue.
  {
    Gemm16Prob& p1 = g.nt(h.Y, H, h.W1, H, P[CAMO_P_F1_B0], nullptr, 0, h.H1, 2 * H, T, 2 * H, H, GF_RELU);
    set_drop(p1, SITE_FFN_RG);
    p1.colmean = w.H1mean; p1.ldm = 2 * H; p1.row_sample = row_sample; p1.inv_nr = inv_nr;
    Gemm16Prob& p2 = g.nt(h.Y2, H, h.W2, H, P[CAMO_P_F2_B0], nullptr, 0, h.H2, 2 * H, TK, 2 * H, H, GF_RELU);
    set_drop(p2, SITE_FFN_KG);
    p2.colmean = w.H2mean; p2.ldm = 2 * H; p2.uniform_n = Nk;
  }
  CK(g.run(), "ffn layer 0");
  return 0;
}

// ---- node-level forward of the fused row-tile schedule: the same function in 3 launches (fused_rows.h) ----
bool fused17_ok(const camo_dims_t& d, const float* const* P, int precision, int Nk, int max_nr) {
  return g_opt_fused != 0 && precision == CAMO_PREC_BF16 && fused17_dims(d) && Nk <= 16 && max_nr <= 64 * FUSED_MAX_SPLITS &&
         P[CAMO_P_RG_PROJ_W] && P[CAMO_P_KG_PROJ_W];
}

// camo_forward_loss_backward's optional caller-owned weight shadows (camo_shadow_bytes): the 14 fragment-order bf16 copies the
// fused kernels stream live there instead of in the per-batch workspace, so that the optimizer call can leave them ready for
// the next step (camo_clip_adamw_shadows) and the forward need not rebuild them (CAMO_FLAG_SHADOWS_VALID).
static thread_local void* t_zero_front_ptr[FUSED_FRONT_MAXZ]; static thread_local unsigned t_zero_front_bytes[FUSED_FRONT_MAXZ];
static thread_local int t_nzero_front = 0;
static thread_local void* t_zero_bwd1_ptr[FUSED_BWD1_MAXZ]; static thread_local unsigned t_zero_bwd1_bytes[FUSED_BWD1_MAXZ];
static thread_local int t_nzero_bwd1 = 0;
static thread_local void* t_shadows = nullptr;
static thread_local bool t_shadows_valid = false;
static thread_local bool t_fold_missing = false;  // camo_forward_cached(shadows_valid = 2): valid shadows that lack the inference calls' folded in-projection
static thread_local int t_shadows_state = 0;    // what the fused forward left in external shadows: 0 untouched, 1 forward set, 2 forward + transposed
struct ShadowSet { us16 *Wrg, *Wkg, *Wqkv_rg, *Wqkv_kg, *Wo1, *Wo2, *W1, *W2, *W1T, *W2T, *Wo1T, *Wo2T, *WcRgT, *WcKgT, *Wf_rg; float* bf_rg; size_t bytes; };
static ShadowSet shadow_carve(void* base) {
  ShadowSet x{};
  Carver c(base);
  const size_t H = 256, D = 128, HH = H * H;
  x.Wrg = c.take<us16>(H * D); x.Wkg = c.take<us16>(H * D); x.Wqkv_rg = c.take<us16>(3 * HH); x.Wqkv_kg = c.take<us16>(3 * HH);
  x.Wo1 = c.take<us16>(HH); x.Wo2 = c.take<us16>(HH); x.W1 = c.take<us16>(2 * HH); x.W2 = c.take<us16>(2 * HH);
  x.W1T = c.take<us16>(2 * HH); x.W2T = c.take<us16>(2 * HH); x.Wo1T = c.take<us16>(HH); x.Wo2T = c.take<us16>(HH);
  x.WcRgT = c.take<us16>(3 * HH); x.WcKgT = c.take<us16>(3 * HH);
  x.Wf_rg = c.take<us16>(3 * H * D); x.bf_rg = c.take<float>(3 * H);      // (inference calls only: launch_fold_rg)
  x.bytes = (c.off + 255) & ~size_t(255);
  return x;
}
static void bind_shadows(Ws& w) {                 // (after every carve() of a call that was handed external shadows)
  if (!t_shadows) return;
  const ShadowSet x = shadow_carve(t_shadows);
  Ws::F17& f = w.f;
  f.Wrg = x.Wrg; f.Wkg = x.Wkg; f.Wqkv_rg = x.Wqkv_rg; f.Wqkv_kg = x.Wqkv_kg; f.Wo1 = x.Wo1; f.Wo2 = x.Wo2; f.W1 = x.W1; f.W2 = x.W2;
  f.W1T = x.W1T; f.W2T = x.W2T; f.Wo1T = x.Wo1T; f.Wo2T = x.Wo2T; f.WcRgT = x.WcRgT; f.WcKgT = x.WcKgT;
  f.Wf_rg = x.Wf_rg; f.bf_rg = x.bf_rg;
}

// Which tile family a fused forward takes (fused_rows.h): 0 = 32-row tiles, one per block of 4 waves (small batches: one tile
// per CU is all there is); 2 / 4 = that many tiles per block of 8 waves (fused_wide.hip), chosen so that the blocks still fill the chip.
static int wide_rt(int T, int max_nr, bool save = false) {
  int rt = g_opt_fused_rt;
  // by size: inference calls from about half a 128-row block per CU on (64-row blocks below that).  Calls that save for a backward stay on the 32-row kernels:
  // the saved-tensor stores of the one-launch kernel are not tuned yet (measured slower: B = 64 step 0.53 vs 0.40 ms)
  // (measured eval forward, us: B = 32 [13.5 k rows] 87 / 85 / 103 for 32-row / 2 / 4 tiles per block; B = 48 [20 k] 116 / 103 / 106; B = 56 [24 k] 131 / 112 / 108)
  if (rt < 0) rt = save ? 0 : (T >= 22528 ? 4 : (T >= 13312 ? 2 : 0));
  if (rt != 0 && rt != 1 && rt != 2 && rt != 4) rt = 0;
  if (rt && max_nr > wide_max_rows(rt) - 64 * rt) rt = 0;
  return rt;
}

// The RG rows' whole forward in one launch of 64-row half-blocks, two independent blocks per CU, + the KG rows' launch behind it
// (fused_wide2.hip).  By size for inference AND training calls (the saving / dropout variants write the backward's saved set); a forced
// fused_rt selects the 8-wave / 32-row kernels.
static bool wide2_taken(int T, int max_nr, bool save, bool dropping) {
  (void)dropping;
  if (g_opt_wide2 == 0 || g_opt_fused_one == 0 || max_nr > wide2_max_rows()) return false;
  if (g_opt_wide2 > 0) return true;
  // training calls from 57 344 rows: their blocks are twice as long (the saved set, the dropout hashes), so the second round of blocks
  // must be nearly full before they beat the 32-row back half (measured, ms per step without / with: B = 96 0.517 / 0.540, B = 128
  // 0.637 / 0.621, B = 192 0.864 / 0.804, B = 256 1.076 / 0.979)
  // inference calls from 10 240 rows (eval forward, us without / with: B = 16 59 / 66, B = 24 72.5 / 69.8, B = 32 77 / 71, B = 48 101 / 82)
  return g_opt_fused_rt < 0 && g_opt_wide_front_rt == 0 && T >= (save ? 57344 : 10240);
}

// Training calls (save): the front half alone on wide blocks -- 64-row blocks from 10 240 packed rows (front 21 -> 17 us at B = 24,
// 31 -> 26 at B = 48, 37 -> 28 at B = 56), 128-row blocks from 28 672.  -> sub-tiles per block, 0 = the 32-row front kernel.
// ONE predicate for everything that rides on that launch (the two-plane tail's weight planes are built by its extra blocks).
static int wide_train_front_rt(int T, int max_nr, bool save) {
  if (!save || wide_rt(T, max_nr, save) || g_opt_fused_rt >= 0 || g_opt_wide_front_rt < 0) return 0;
  if (T < 10240 && g_opt_wide_front_rt <= 0) return 0;
  const int wf_rt = g_opt_wide_front_rt > 0 ? g_opt_wide_front_rt : (T >= 4 * 32 * 224 ? 4 : 2);
  if (wf_rt != 1 && wf_rt != 2 && wf_rt != 4) return 0;
  return max_nr <= wide_max_rows(wf_rt) - 64 * wf_rt ? wf_rt : 0;
}

static bool tailw_taken(const camo_dims_t& d, int B, int T, int max_nr, bool save, bool dropping) {
  // (B <= 32: the grouped fp32 tail, forward only, is the shorter one: 29.8 vs 33.5 us at B = 32; equal at 48)
  return g_opt_tailw != 0 && g_opt_fused_one != 0 && (wide_rt(T, max_nr, save) >= 2 || wide2_taken(T, max_nr, save, dropping)) && tail_wide_ok(B, d.num_classes) &&
         (g_opt_tailw > 0 || B > 32 || !tail_fused_ok(B, d.num_classes));
}

int forward_nodes17(const camo_dims_t& d, const float* const* P, const float* rg, const int32_t* rg_offsets, const Desc& bd,
                    const float* kg, int B, int T, int Nk, int max_nr, const Ws& w, const DropCfg& drop, bool save, hipStream_t st, int want_tailw = 0) {
  const int H = 256, D = 128, TK = B * Nk;
  const size_t HH = (size_t)H * H;
  const Ws::F17& f = w.f;
  {   // weight shadows (bf16, fragment order) + the clear of the step's atomics block
    ShadowBatch sb; std::memset(&sb, 0, sizeof(sb));
    auto job = [&](us* dst, int N, int K, const float* s0, int r0, const float* s1 = nullptr, int r1 = 0) {
      ShadowJob& J = sb.j[sb.n++];
      J.dst = dst; J.N = N; J.K = K; J.transposed = 0; J.nsrc = s1 ? 2 : 1;
      J.src[0] = s0; J.rows[0] = r0; J.ld[0] = K; J.src[1] = s1; J.rows[1] = r1; J.ld[1] = K;
    };
    const bool build = !t_shadows_valid;        // (valid: camo_clip_adamw_shadows left them ready; only the clears ride in this launch)
    if (t_shadows) t_shadows_state = save ? 2 : 1;
    if (build) {
    job(f.Wrg, H, D, P[CAMO_P_RG_PROJ_W], H); job(f.Wkg, H, D, P[CAMO_P_KG_PROJ_W], H);
    job(f.Wqkv_rg, 3 * H, H, P[CAMO_P_A1_IN_W], H, P[CAMO_P_A2_IN_W] + HH, 2 * H);       // [Wq1; Wk2; Wv2]: what RG rows are projected with
    job(f.Wqkv_kg, 3 * H, H, P[CAMO_P_A2_IN_W], H, P[CAMO_P_A1_IN_W] + HH, 2 * H);       // [Wq2; Wk1; Wv1]
    job(f.Wo1, H, H, P[CAMO_P_A1_OUT_W], H); job(f.Wo2, H, H, P[CAMO_P_A2_OUT_W], H);
    job(f.W1, 2 * H, H, P[CAMO_P_F1_W0], 2 * H); job(f.W2, 2 * H, H, P[CAMO_P_F2_W0], 2 * H);
    }
    auto zero = [&](void* ptr, size_t bytes) { if (bytes) { sb.zero_ptr[sb.nzero] = ptr; sb.zero_bytes[sb.nzero++] = (bytes + 15) & ~size_t(15); } };
    if (save) {
      // transposed shadows of the backward's dy . W products; K-concatenated where one product serves three in-projections
      auto jobT = [&](us* dst, int N, int K, const float* s0, int r0, const float* s1 = nullptr, int r1 = 0) {
        ShadowJob& J = sb.j[sb.n++];
        J.dst = dst; J.N = N; J.K = K; J.transposed = 1; J.nsrc = s1 ? 2 : 1;
        J.src[0] = s0; J.rows[0] = r0; J.ld[0] = N; J.src[1] = s1; J.rows[1] = r1; J.ld[1] = N;
      };
      if (build) {
      jobT(f.W1T, H, 2 * H, P[CAMO_P_F1_W0], 2 * H); jobT(f.W2T, H, 2 * H, P[CAMO_P_F2_W0], 2 * H);
      jobT(f.Wo1T, H, H, P[CAMO_P_A1_OUT_W], H); jobT(f.Wo2T, H, H, P[CAMO_P_A2_OUT_W], H);
      jobT(f.WcRgT, H, 3 * H, P[CAMO_P_A1_IN_W], H, P[CAMO_P_A2_IN_W] + HH, 2 * H);      // dR = [dQ | dK2 | dV2] . [Wq1; Wk2; Wv2]
      jobT(f.WcKgT, H, 3 * H, P[CAMO_P_A2_IN_W], H, P[CAMO_P_A1_IN_W] + HH, 2 * H);      // dG = [dQ2 | dK | dV] . [Wq2; Wk1; Wv1]
      }
      zero(w.zero_base, w.zero_bytes);                       // means, dfused, arrival counters, dK|dV and dQ2 sums
      // pad rows (row count rounded up to 128) of every weight-gradient operand: the contraction runs over whole 64-row tiles
      const size_t t = T, tk = TK;
      auto pad = [&](us* buf, size_t rows, size_t rows_p, size_t width) { zero(buf + rows * width, (rows_p - rows) * width * sizeof(us)); };
      pad(f.X16, t, w.Tp, D); pad(f.R16, t, w.Tp, H); pad(f.O16, t, w.Tp, H); pad(f.Y16, t, w.Tp, H); pad(f.dH16, t, w.Tp, 2 * H);
      pad(f.dU16, t, w.Tp, H); pad(f.dQKV16, t, w.Tp, 3 * H); pad(f.dR16, t, w.Tp, H);
      pad(f.KG16, tk, w.TKp, D); pad(f.G16, tk, w.TKp, H); pad(f.O2_16, tk, w.TKp, H); pad(f.Y2_16, tk, w.TKp, H); pad(f.dH2_16, tk, w.TKp, 2 * H);
      pad(f.dU2_16, tk, w.TKp, H); pad(f.dQKVkg16, tk, w.TKp, 3 * H); pad(f.dG16, tk, w.TKp, H);
      zero(w.dHm1, (size_t)B * 2 * H * sizeof(float)); zero(w.dHm2, (size_t)B * 2 * H * sizeof(float));   // atomically summed by the one-launch tail
    } else {
      zero(w.zero_base, (size_t)(reinterpret_cast<char*>(w.dKV) - reinterpret_cast<char*>(w.zero_base)));
      // (the one-launch tail's all-reduce buffers and counter words; the parameter-space block behind them belongs to the backward)
      zero(w.tailsum, ((size_t)B * 4 * H + ((tail_counter_words(B) + 3) & ~size_t(3))) * sizeof(float));
    }
    // inference calls: the RG rows' folded in-projection rides with every rebuild of the forward set, and alone when the caller's valid
    // shadows come from the optimizer call, which does not build it (camo_forward_cached, shadows_valid = 2)
    const bool fold = !save && (build || t_fold_missing);
    if (build) {
      CK(launch_weight_shadows(sb, st), "weight shadows");
      if (fold) CK(launch_fold_rg(P[CAMO_P_A1_IN_W], P[CAMO_P_A2_IN_W] + HH, P[CAMO_P_A1_IN_B], P[CAMO_P_A2_IN_B] + H, P[CAMO_P_RG_PROJ_W], P[CAMO_P_RG_PROJ_B],
                                   f.Wf_rg, f.bf_rg, st), "folded in-projection");
      t_nzero_front = t_nzero_bwd1 = 0;
    } else {
      if (fold) CK(launch_fold_rg(P[CAMO_P_A1_IN_W], P[CAMO_P_A2_IN_W] + HH, P[CAMO_P_A1_IN_B], P[CAMO_P_A2_IN_B] + H, P[CAMO_P_RG_PROJ_W], P[CAMO_P_RG_PROJ_B],
                                  f.Wf_rg, f.bf_rg, st), "folded in-projection");
      // no shadow launch this step: the clears ride elsewhere -- the atomics block and d(mean H) at the end of the front
      // kernel's blocks (first use: the back kernel's pooled sums), the operand pad rows in extra blocks of the first backward
      // kernel (first use: the weight-gradient launch)
      t_nzero_front = t_nzero_bwd1 = 0;
      for (int i = 0; i < sb.nzero; ++i) {
        const bool early = sb.zero_ptr[i] == (void*)w.zero_base || sb.zero_ptr[i] == (void*)w.dHm1 || sb.zero_ptr[i] == (void*)w.dHm2 || sb.zero_ptr[i] == (void*)w.tailsum;
        if (sb.zero_bytes[i] > 0xFFFFFFF0ull) return fail(CAMO_E_ARG, "clear range too large");
        if (early) {
          if (t_nzero_front >= FUSED_FRONT_MAXZ) return fail(CAMO_E_ARG, "too many early clear ranges");
          t_zero_front_ptr[t_nzero_front] = sb.zero_ptr[i]; t_zero_front_bytes[t_nzero_front++] = (unsigned)sb.zero_bytes[i];
        } else {
          if (t_nzero_bwd1 >= FUSED_BWD1_MAXZ) return fail(CAMO_E_ARG, "too many late clear ranges");
          t_zero_bwd1_ptr[t_nzero_bwd1] = sb.zero_ptr[i]; t_zero_bwd1_bytes[t_nzero_bwd1++] = (unsigned)sb.zero_bytes[i];
        }
      }
    }
  }
  FrontArgs fa; std::memset(&fa, 0, sizeof(fa));
  fa.qscale = 1.0f / sqrtf(32.0f); fa.save = save ? 1 : 0;
  fa.s[0] = FrontStream{rg, T, f.Wrg, P[CAMO_P_RG_PROJ_B], f.Wqkv_rg, P[CAMO_P_A1_IN_B], P[CAMO_P_A2_IN_B] + H, f.X16, f.R16, f.Q16, f.KV2_16, 0};
  fa.s[1] = FrontStream{kg, TK, f.Wkg, P[CAMO_P_KG_PROJ_B], f.Wqkv_kg, P[CAMO_P_A2_IN_B], P[CAMO_P_A1_IN_B] + H, f.KG16, f.G16, f.Q2_16, f.KV16, 0};
  fa.stamps = g_dbg_stamps; fa.exp = g_opt_exp;
  fa.nzero = t_nzero_front;
  for (int i = 0; i < t_nzero_front; ++i) { fa.zero_ptr[i] = t_zero_front_ptr[i]; fa.zero_bytes[i] = t_zero_front_bytes[i]; }
  t_nzero_front = 0;
  const bool w2 = wide2_taken(T, max_nr, save, drop.p > 0.f);
  const int rt = w2 ? 0 : wide_rt(T, max_nr, save);
  const bool one = w2 || (rt >= 2 && g_opt_fused_one != 0);
  const int wf_rt = wide_train_front_rt(T, max_nr, save);
  const bool wide_train_front = wf_rt != 0;
  if (want_tailw && !(one || wide_train_front)) return fail(CAMO_E_ARG, "two-plane tail asked for on a call whose front launch cannot build its weight planes");
  if ((one || wide_train_front) && want_tailw) {
    // the per-sample tail's weights as hi / lo bf16 planes in fragment order: extra blocks of the (KG rows') wide front launch
    const size_t HH = (size_t)H * H;
    const float* hsrc[4] = {P[CAMO_P_HEADS], P[CAMO_P_HEADS + 4], P[CAMO_P_HEADS + 8], P[CAMO_P_HEADS + 12]};
    auto xj = [&](us* dst, int N, int K, const float* src, int lo) {
      ShadowJob& J = fa.xjob[fa.nxjob++];
      std::memset(&J, 0, sizeof(J));
      J.dst = dst; J.N = N; J.K = K; J.nsrc = 1; J.src[0] = src; J.rows[0] = N; J.ld[0] = K; J.lo = lo;
    };
    (void)HH;
    for (int lo = 0; lo < 2; ++lo) {
      xj(w.f.tailw[0 + lo], H, 2 * H, P[CAMO_P_F1_W3], lo); xj(w.f.tailw[2 + lo], H, 2 * H, P[CAMO_P_F2_W3], lo);
      xj(w.f.tailw[4 + lo], H, 2 * H, P[CAMO_P_FU_W0], lo); xj(w.f.tailw[6 + lo], H, H, P[CAMO_P_FU_W3], lo);
    }
    for (int lo = 0; lo < 2; ++lo) {      // the four heads' first layers [128 x 256] each, stacked
      ShadowJob& J = fa.xjob[fa.nxjob++];
      std::memset(&J, 0, sizeof(J));
      J.dst = w.f.tailw[8 + lo]; J.N = 2 * H; J.K = H; J.nsrc = 4; J.lo = lo;
      for (int x = 0; x < 4; ++x) { J.src[x] = hsrc[x]; J.rows[x] = H / 2; J.ld[x] = H; }
    }
    if (want_tailw > 1) {                   // the transposed planes of the tail's backward (tail_wide.h, TailWideBwdArgs)
      auto xjT = [&](us* dst, int N, int K, const float* src, int ld, int lo) {      // shadow of src^T: src has K rows of >= N columns
        ShadowJob& J = fa.xjob[fa.nxjob++];
        std::memset(&J, 0, sizeof(J));
        J.dst = dst; J.N = N; J.K = K; J.transposed = 1; J.nsrc = 1; J.src[0] = src; J.rows[0] = K; J.ld[0] = ld; J.lo = lo;
      };
      for (int lo = 0; lo < 2; ++lo) {
        ShadowJob& J = fa.xjob[fa.nxjob++];   // [Wh0_0; ..; Wh0_3]^T: 512 source rows (4 x 128) of 256 columns
        std::memset(&J, 0, sizeof(J));
        J.dst = w.f.tailw[10 + lo]; J.N = H; J.K = 2 * H; J.transposed = 1; J.nsrc = 4; J.lo = lo;
        for (int x = 0; x < 4; ++x) { J.src[x] = hsrc[x]; J.rows[x] = H / 2; J.ld[x] = H; }
        xjT(w.f.tailw[12 + lo], H, H, P[CAMO_P_FU_W3], H, lo);
        xjT(w.f.tailw[14 + lo], 2 * H, H, P[CAMO_P_FU_W0], 2 * H, lo);
        xjT(w.f.tailw[16 + lo], 2 * H, H, P[CAMO_P_F1_W3], 2 * H, lo);
        xjT(w.f.tailw[18 + lo], 2 * H, H, P[CAMO_P_F2_W3], 2 * H, lo);
      }
    }
  }
  if (one) { fa.split3 = 1; CK(launch_wide_front(fa, 1, st, 1), "fused forward, KG rows' front half (32-row tiles, one in-projection pass per block)"); }
  else if (rt) CK(launch_wide_front(fa, rt, st, 0), "fused forward, front half (wide tiles)");
  // (the back half of training calls stays on the 32-row kernel, whose saving + dropout variant is the faster one: 77 vs 94 us at B = 64)
  else if (wide_train_front)
    CK(launch_wide_front(fa, wf_rt, st, 0), "fused forward, front half (wide tiles)");
  else CK(launch_fused_front(fa, g_opt_fused_variant, st), "fused forward, front half");
  BackArgs ba; std::memset(&ba, 0, sizeof(ba));
  ba.s[0] = BackStream{f.Wo1, P[CAMO_P_A1_OUT_B], f.W1, P[CAMO_P_F1_B0], P[CAMO_P_LN1_W], P[CAMO_P_LN1_B], f.R16,
                       f.O16, f.Y16, f.XH16, f.rstd1, f.mask1, w.Ymean, w.H1mean, SITE_FFN_RG};
  ba.s[1] = BackStream{f.Wo2, P[CAMO_P_A2_OUT_B], f.W2, P[CAMO_P_F2_B0], P[CAMO_P_LN2_W], P[CAMO_P_LN2_B], f.G16,
                       f.O2_16, f.Y2_16, f.XH2_16, f.rstd2, f.mask2, w.Y2mean, w.H2mean, SITE_FFN_KG};
  ba.Q16 = f.Q16; ba.KV16 = f.KV16; ba.Q2_16 = f.Q2_16; ba.KV2_16 = f.KV2_16;
  ba.off = rg_offsets; ba.tile_off = bd.tile_off; ba.tile_desc = bd.tile_desc; ba.inv_nr = bd.inv_nr; ba.lse2 = f.lse2;
  ba.B = B; ba.Nk = Nk; ba.rows_rg = T; ba.rg_tiles_max = T / 32 + B;          // >= sum of ceil(Nr / 32); surplus blocks exit at once
  ba.part = f.part; ba.tickets = w.tickets; ba.max_splits = (max_nr + 63) / 64;
  ba.drop = drop; ba.save = save ? 1 : 0; ba.exp = g_opt_exp;
  ba.stamps = g_dbg_stamps ? g_dbg_stamps + (size_t)g_dbg_stamp_blocks * 8 : nullptr;
  // (R16 is read by the row-space form of the projections' weight gradients only: the same predicate as backward_nodes17's; tests that
  // read it back run an inference call with fused_save)
  const bool param_space_bwd = g_opt_param_space < 0 ? T >= 10240 : g_opt_param_space > 0;
  if (w2) CK(launch_wide2_rgfwd(fa.s[0], f.Wf_rg, f.bf_rg, fa.qscale, ba, max_nr, (!param_space_bwd || g_opt_fused_save != 0) ? 1 : 0, st),
             "fused forward, RG rows in one launch (64-row half-blocks)");
  else if (one) CK(launch_wide_rgfwd(fa.s[0], fa.qscale, ba, rt, max_nr, st), "fused forward, RG rows in one launch (wide tiles)");
  else if (rt) CK(launch_wide_back(ba, rt, max_nr, st), "fused forward, back half (wide tiles)");
  else CK(launch_fused_back(ba, g_opt_fused_variant, st), "fused forward, back half");
  return 0;
}

// ---- node-level backward of the fused row-tile schedule (w.dcomb, w.dHm1, w.dHm2 hold the pooled gradients) ----
int backward_nodes17(const camo_dims_t& d, const float* const* P, float* const* Gr, const int32_t* rg_offsets, const Desc& bd,
                     int B, int T, int Nk, const Ws& w, const DropCfg& drop, hipStream_t st);
int tail17(const camo_dims_t& d, const float* const* P, float* const* Gr, const Ws& w, int B, float* outs, const FusedLoss* fl,
           const DropCfg& drop, hipStream_t st);

// ---- node-level backward of the bf16 schedule (w.dcomb, w.dHm1, w.dHm2 hold the pooled gradients) ----
int backward_nodes16(const camo_dims_t& d, const float* const* P, float* const* Gr, const int32_t* rg_offsets,
                     const int32_t* row_sample, const float* inv_nr, int B, int T, int Nk, int max_nr, const Ws& w,
                     const DropCfg& drop, hipStream_t st) {
  const int H = d.hidden_dim, D = d.rg_dim, Dk = d.kg_dim, TK = B * Nk, nh = d.num_heads;
  const size_t HH = (size_t)H * H;
  const Ws::H16& h = w.h;
  // dH1 = mask(H1) * bcast(dHm1) / n is needed only as a GEMM operand.  At hidden_dim 256 and B <= 32 the whole-row
  // kernel builds it while staging (GF_A_VIRT) from the bf16 activation mask and the per-sample gradient rows, for both
  // products that read it; otherwise relu_bcast_bwd writes it out.
  const bool virt = H == 256 && B <= 32;
  if (!virt) {
    BcastSeg s0{nullptr, w.dHm1, 2 * H, row_sample, inv_nr, 0, nullptr, T, h.dH1, h.H1};
    BcastSeg s1{nullptr, w.dHm2, 2 * H, nullptr, nullptr, Nk, nullptr, TK, h.dH2, h.H2};
    CK(launch_relu_bcast_bwd(s0, s1, 2 * H, drop.scale, st), "relu bcast bwd");
  }
  auto make_virt = [&](Gemm16Prob& p, const float* g_rows, bool rg_side) {
    p.flags |= GF_A_VIRT; p.virt_g = g_rows; p.ldg = 2 * H; p.aux_scale = drop.scale;
    if (rg_side) { p.row_sample = row_sample; p.inv_nr = inv_nr; p.uniform_n = 0; } else { p.row_sample = nullptr; p.inv_nr = nullptr; p.uniform_n = Nk; }
    if (p.flags & GF_A_KMAJOR) p.ldc16 = B;               // (sample count of a weight-gradient problem)
  };
  GB16 g(drop, st);
  // first FFN layer: dY = bcast(dpool)/n + dH1.W1 ; dW1 += dH1^T.Y -- and, at hidden_dim 256, the LayerNorm backward
  // dY -> dU (+ dgamma, dbeta) as the epilogue of the dY product (whole-row tiles)
  if (H == 256) {
    Gemm16Prob& p1 = g.nt(virt ? h.H1 : h.dH1, 2 * H, h.W1T, 2 * H, nullptr, w.dU, H, h.dU, H, T, H, 2 * H);
    set_bcast(p1, w.dcomb, 2 * H, row_sample, inv_nr, 0);
    p1.ln_mode = 2; p1.ln_gamma = P[CAMO_P_LN1_W]; p1.ln_stats = w.st1; p1.ln_x = w.U;
    p1.ln_dgamma = Gr[CAMO_P_LN1_W]; p1.ln_dbeta = Gr[CAMO_P_LN1_B];
    if (virt) make_virt(p1, w.dHm1, true);
    Gemm16Prob& p2 = g.nt(virt ? h.H2 : h.dH2, 2 * H, h.W2T, 2 * H, nullptr, w.dU2, H, h.dU2, H, TK, H, 2 * H);
    set_bcast(p2, w.dcomb + H, 2 * H, nullptr, nullptr, Nk);
    p2.ln_mode = 2; p2.ln_gamma = P[CAMO_P_LN2_W]; p2.ln_stats = w.st2; p2.ln_x = w.U2;
    p2.ln_dgamma = Gr[CAMO_P_LN2_W]; p2.ln_dbeta = Gr[CAMO_P_LN2_B];
    if (virt) make_virt(p2, w.dHm2, false);
  } else {
    set_bcast(g.nt(h.dH1, 2 * H, h.W1T, 2 * H, nullptr, w.dY, H, nullptr, 0, T, H, 2 * H), w.dcomb, 2 * H, row_sample, inv_nr, 0);
    set_bcast(g.nt(h.dH2, 2 * H, h.W2T, 2 * H, nullptr, w.dY2, H, nullptr, 0, TK, H, 2 * H), w.dcomb + H, 2 * H, nullptr, nullptr, Nk);
  }
  {
    Gemm16Prob& t1 = g.tn(virt ? h.H1 : h.dH1, 2 * H, h.Y, H, Gr[CAMO_P_F1_W0], H, Gr[CAMO_P_F1_B0], 2 * H, H, T);
    if (virt) make_virt(t1, w.dHm1, true);
    Gemm16Prob& t2 = g.tn(virt ? h.H2 : h.dH2, 2 * H, h.Y2, H, Gr[CAMO_P_F2_W0], H, Gr[CAMO_P_F2_B0], 2 * H, H, TK);
    if (virt) make_virt(t2, w.dHm2, false);
  }
  CK(g.run(), "ffn layer 0 bwd");
  if (H != 256) {
    LnBwdSeg s0{w.U, w.dY, w.st1, P[CAMO_P_LN1_W], w.dU, Gr[CAMO_P_LN1_W], Gr[CAMO_P_LN1_B], T, h.dU};
    LnBwdSeg s1{w.U2, w.dY2, w.st2, P[CAMO_P_LN2_W], w.dU2, Gr[CAMO_P_LN2_W], Gr[CAMO_P_LN2_B], TK, h.dU2};
    CK(launch_ln_bwd(s0, s1, H, st), "layernorm bwd");
  }
  // out-projections
  g.nt(h.dU, H, h.Wo1T, H, nullptr, w.dO, H, nullptr, 0, T, H, H);
  g.nt(h.dU2, H, h.Wo2T, H, nullptr, w.dO2, H, nullptr, 0, TK, H, H);
  g.tn(h.dU, H, h.O, H, Gr[CAMO_P_A1_OUT_W], H, Gr[CAMO_P_A1_OUT_B], H, H, T);
  g.tn(h.dU2, H, h.O2, H, Gr[CAMO_P_A2_OUT_W], H, Gr[CAMO_P_A2_OUT_B], H, H, TK);
  CK(g.run(), "out-projection bwd");
  // attention cores.  Their node-side outputs are GEMM operands only, so they are written as bf16 straight into the
  // concatenated [dQ | dK2 | dV2] (rg rows) and [dQ2 | dK | dV] (kg rows) operands of the in-projection backward.
  // Both directions run in one launch, each (head, sample) owned by one block, so nothing is accumulated with atomics.
  CK(launch_attn_bwd_pair(w.Q, w.KV, w.P, w.dO, w.Q2, w.KV2, w.P2, w.dO2, rg_offsets, Bf16Dst{h.dQKV, 3 * H},
                          Bf16Dst{h.dQKVkg + H, 3 * H}, Bf16Dst{h.dQKVkg, 3 * H}, Bf16Dst{h.dQKV + H, 3 * H}, w.O2, B, H, nh, Nk, drop, st),
     "attention bwd (both directions)");
  // in-projections: one K = 3H product per side for the input gradient (dR = dU + [dQ|dK2|dV2].[Wq1;Wk2;Wv2]),
  // and the four weight gradients
  set_res(g.nt(h.dQKV, 3 * H, h.WcRgT, 3 * H, nullptr, nullptr, 0, h.dR, H, T, H, 3 * H), w.dU, H);
  set_res(g.nt(h.dQKVkg, 3 * H, h.WcKgT, 3 * H, nullptr, nullptr, 0, h.dG, H, TK, H, 3 * H), w.dU2, H);
  CK(g.run(), "in-projection bwd (input gradients)");
  // every remaining weight gradient in one launch: the in-projection ones do not wait for dR / dG, but beside the
  // K = 3H products above they set that launch's length, while the lone dW_rg / dW_kg launch left most CUs idle
  g.tn(h.dQKV, 3 * H, h.R, H, Gr[CAMO_P_A1_IN_W], H, Gr[CAMO_P_A1_IN_B], H, H, T);
  g.tn(h.dQKV + H, 3 * H, h.R, H, Gr[CAMO_P_A2_IN_W] + HH, H, Gr[CAMO_P_A2_IN_B] + H, 2 * H, H, T);
  g.tn(h.dQKVkg, 3 * H, h.G, H, Gr[CAMO_P_A2_IN_W], H, Gr[CAMO_P_A2_IN_B], H, H, TK);
  g.tn(h.dQKVkg + H, 3 * H, h.G, H, Gr[CAMO_P_A1_IN_W] + HH, H, Gr[CAMO_P_A1_IN_B] + H, 2 * H, H, TK);
  g.tn(h.dR, H, h.X, D, Gr[CAMO_P_RG_PROJ_W], D, Gr[CAMO_P_RG_PROJ_B], H, D, T);
  g.tn(h.dG, H, h.KG, Dk, Gr[CAMO_P_KG_PROJ_W], Dk, Gr[CAMO_P_KG_PROJ_B], H, Dk, TK);
  CK(g.run(), "in-projection and input-projection weight gradients");
  return 0;
}

// the per-sample tail of the fused schedule as one launch (misc.hip, tail_fused_kernel); fl == null: forward only
int tail17(const camo_dims_t& d, const float* const* P, float* const* Gr, const Ws& w, int B, float* outs, const FusedLoss* fl,
           const DropCfg& drop, hipStream_t st) {
  const int H = 256;
  TailFusedArgs a; std::memset(&a, 0, sizeof(a));
  a.Ymean = w.Ymean; a.H1mean = w.H1mean; a.Y2mean = w.Y2mean; a.H2mean = w.H2mean;
  a.W13 = P[CAMO_P_F1_W3]; a.b13 = P[CAMO_P_F1_B3]; a.W23 = P[CAMO_P_F2_W3]; a.b23 = P[CAMO_P_F2_B3];
  a.Wfu0 = P[CAMO_P_FU_W0]; a.bfu0 = P[CAMO_P_FU_B0]; a.Wfu3 = P[CAMO_P_FU_W3]; a.bfu3 = P[CAMO_P_FU_B3];
  for (int x = 0; x < 4; ++x) {
    a.Wh0[x] = P[CAMO_P_HEADS + 4 * x]; a.bh0[x] = P[CAMO_P_HEADS + 4 * x + 1]; a.Wh3[x] = P[CAMO_P_HEADS + 4 * x + 2]; a.bh3[x] = P[CAMO_P_HEADS + 4 * x + 3];
  }
  if (fl) {
    a.gW13 = Gr[CAMO_P_F1_W3]; a.gb13 = Gr[CAMO_P_F1_B3]; a.gW23 = Gr[CAMO_P_F2_W3]; a.gb23 = Gr[CAMO_P_F2_B3];
    a.gWfu0 = Gr[CAMO_P_FU_W0]; a.gbfu0 = Gr[CAMO_P_FU_B0]; a.gWfu3 = Gr[CAMO_P_FU_W3]; a.gbfu3 = Gr[CAMO_P_FU_B3];
    for (int x = 0; x < 4; ++x) {
      a.gWh0[x] = Gr[CAMO_P_HEADS + 4 * x]; a.gbh0[x] = Gr[CAMO_P_HEADS + 4 * x + 1]; a.gWh3[x] = Gr[CAMO_P_HEADS + 4 * x + 2]; a.gbh3[x] = Gr[CAMO_P_HEADS + 4 * x + 3];
    }
    a.y = reinterpret_cast<const long long*>(fl->y); a.e = fl->e; a.s = fl->s; a.terms = fl->terms; a.pred = fl->pred;
    a.dcomb = w.dcomb; a.dHm1 = w.dHm1; a.dHm2 = w.dHm2;
  }
  a.outs = outs;
  a.F1sum = w.tailsum; a.hidsum = w.tailsum + (size_t)B * H; a.dF1sum = w.tailsum + (size_t)B * 3 * H;
  a.counters = reinterpret_cast<unsigned int*>(w.tailsum + (size_t)B * 4 * H);
  a.B = B; a.C = d.num_classes; a.mode = fl ? 1 : 0; a.drop = drop;
  a.stamps = g_dbg_stamps ? g_dbg_stamps + (size_t)4 * g_dbg_stamp_blocks * 8 : nullptr;
  const bool groups = fl && B > 16;
  if (groups) { a.comb_out = w.comb; a.F1_out = w.F1; a.fused_out = w.fused; a.dhid_out = w.dhid; a.dfused_out = w.dfused; a.dF1_out = w.dF1; }
  CK(launch_tail_fused(a, st), "per-sample tail (one launch)");
  if (groups) {
    // more than one group of 16 samples: the big weight gradients of the tail are sums over every group -- one batched launch
    // (contraction over the B samples), operands = the copies the tail kernel left in the workspace
    const int Fh = H / 2;
    GB g(drop, 0, st);
    for (int x = 0; x < 4; ++x) g.tn(w.dhid + x * Fh, 4 * Fh, w.fused, H, a.gWh0[x], H, a.gbh0[x], Fh, H, B);
    g.tn(w.dfused, H, w.F1, H, a.gWfu3, H, a.gbfu3, H, H, B);
    g.tn(w.dF1, H, w.comb, 2 * H, a.gWfu0, 2 * H, a.gbfu0, H, 2 * H, B);
    g.tn(w.dcomb, 2 * H, w.H1mean, 2 * H, a.gW13, 2 * H, a.gb13, H, 2 * H, B);
    g.tn(w.dcomb + H, 2 * H, w.H2mean, 2 * H, a.gW23, 2 * H, a.gb23, H, 2 * H, B);
    CK(g.run(), "per-sample tail, weight gradients");
  }
  return 0;
}

int backward_nodes17(const camo_dims_t& d, const float* const* P, float* const* Gr, const int32_t* rg_offsets, const Desc& bd,
                     int B, int T, int Nk, const Ws& w, const DropCfg& drop, hipStream_t st) {
  const int H = 256, D = 128, TK = B * Nk;
  const size_t HH = (size_t)H * H;
  const Ws::F17& f = w.f;
  // The projections' and in-projections' weight gradients in parameter space (no dR / dG product in the second kernel, 131 k instead
  // of 427 k MACs per row, one small launch behind the weight gradients): from ~10 k packed rows on -- below that the extra launch
  // (~10 us) costs what the second kernel saves (measured: B = 16 +10 us, B = 64 -22 us, B = 256 -105 us per step)
  const bool param_space = g_opt_param_space < 0 ? T >= 10240 : g_opt_param_space > 0;
  Bwd1Args a1; std::memset(&a1, 0, sizeof(a1));
  a1.s[0] = Bwd1Stream{f.W1T, f.Wo1T, f.mask1, f.XH16, f.rstd1, P[CAMO_P_LN1_W], w.dHm1, 2 * H, w.dcomb, 2 * H, f.dH16, f.dU16,
                       Gr[CAMO_P_LN1_W], Gr[CAMO_P_LN1_B]};
  a1.s[1] = Bwd1Stream{f.W2T, f.Wo2T, f.mask2, f.XH2_16, f.rstd2, P[CAMO_P_LN2_W], w.dHm2, 2 * H, w.dcomb + H, 2 * H, f.dH2_16, f.dU2_16,
                       Gr[CAMO_P_LN2_W], Gr[CAMO_P_LN2_B]};
  a1.Q16 = f.Q16; a1.KV16 = f.KV16; a1.dQKV16 = f.dQKV16; a1.dKV = w.dKV;
  a1.O2_16 = f.O2_16; a1.dO2_16 = f.dO2_16; a1.delta2 = f.delta2;
  a1.off = rg_offsets; a1.tile_off = bd.tile_off; a1.tile_desc = bd.tile_desc; a1.inv_nr = bd.inv_nr; a1.row_sample = bd.row_sample;
  a1.B = B; a1.Nk = Nk; a1.rows_rg = T; a1.rg_tiles_max = T / 32 + B; a1.qscale = 1.0f / sqrtf(32.0f); a1.drop = drop;
  a1.stamps = g_dbg_stamps ? g_dbg_stamps + (size_t)2 * g_dbg_stamp_blocks * 8 : nullptr;
  a1.nzero = t_nzero_bwd1; a1.exp = g_opt_exp;
  for (int i = 0; i < t_nzero_bwd1; ++i) { a1.zero_ptr[i] = t_zero_bwd1_ptr[i]; a1.zero_bytes[i] = t_zero_bwd1_bytes[i]; }
  t_nzero_bwd1 = 0;
  // the RG rows of the first half on 64-row half-blocks (bwd_wide2.hip) from 16 384 packed rows (training step, ms without / with:
  // B = 24 0.209 / 0.217, B = 32 0.2395 / 0.237, B = 48 0.307 / 0.301, B = 64 0.355 / 0.349, B = 128 0.632 / 0.615, B = 256 1.007 / 0.944,
  // B = 1024 3.215 / 2.815) -- behind either forward: the saved set is the same
  const bool bwd1w = g_opt_wide2_bwd != 0 && (g_opt_wide2_bwd > 0 || (g_opt_fused_rt < 0 && T >= 16384));
  if (bwd1w) CK(launch_wide2_bwd1(a1, g_opt_fused_variant, st), "fused backward, first half (64-row half-blocks)");
  else       CK(launch_fused_bwd1(a1, g_opt_fused_variant, st), "fused backward, first half");
  Bwd2Args a2; std::memset(&a2, 0, sizeof(a2));
  a2.Q2_16 = f.Q2_16; a2.dO2_16 = f.dO2_16; a2.lse2 = f.lse2; a2.delta2 = f.delta2; a2.KV2_16 = f.KV2_16; a2.dQKV16 = f.dQKV16;
  a2.dU16 = f.dU16; a2.WcRgT = f.WcRgT; a2.dR16 = f.dR16; a2.dQ2acc = w.dQ2acc; a2.dKV = w.dKV;
  a2.dU2_16 = f.dU2_16; a2.WcKgT = f.WcKgT; a2.dQKVkg16 = f.dQKVkg16; a2.dG16 = f.dG16; a2.dGpart = f.dGpart;
  a2.tickets = w.tickets + B; a2.off = rg_offsets; a2.tile_off = bd.tile_off; a2.tile_desc = bd.tile_desc;
  a2.B = B; a2.Nk = Nk; a2.rows_rg = T; a2.rg_tiles_max = T / 32 + B; a2.qscale = a1.qscale; a2.drop = drop;
  a2.stamps = g_dbg_stamps ? g_dbg_stamps + (size_t)3 * g_dbg_stamp_blocks * 8 : nullptr;
  a2.param_space = param_space ? 1 : 0;
  a2.split_finish = (param_space && bwd1w) ? 1 : 0;        // (by the size rule of the wide first half: the extra launch costs ~2 us)
  // (wide2_bwd == 2: developer A/B, bwd2p_kernel + bwd2_finish_kernel behind the wide first half)
  if (a2.split_finish && g_opt_wide2_bwd != 2) CK(launch_wide2_bwd2(a2, st), "fused backward, second half (64-row blocks)");
  else CK(launch_fused_bwd2(a2, g_opt_fused_variant, st), "fused backward, second half");
  // every node-level weight gradient: dW += dy^T . x over the rows of a stream (bf16 operands the fused kernels wrote)
  if (!param_space) {
    GB16 g(drop, st);
    g.tn(f.dH16, 2 * H, f.Y16, H, Gr[CAMO_P_F1_W0], H, Gr[CAMO_P_F1_B0], 2 * H, H, T);
    g.tn(f.dU16, H, f.O16, H, Gr[CAMO_P_A1_OUT_W], H, Gr[CAMO_P_A1_OUT_B], H, H, T);
    g.tn(f.dQKV16, 3 * H, f.R16, H, Gr[CAMO_P_A1_IN_W], H, Gr[CAMO_P_A1_IN_B], H, H, T);
    g.tn(f.dQKV16 + H, 3 * H, f.R16, H, Gr[CAMO_P_A2_IN_W] + HH, H, Gr[CAMO_P_A2_IN_B] + H, 2 * H, H, T);
    g.tn(f.dR16, H, f.X16, D, Gr[CAMO_P_RG_PROJ_W], D, Gr[CAMO_P_RG_PROJ_B], H, D, T);
    g.tn(f.dH2_16, 2 * H, f.Y2_16, H, Gr[CAMO_P_F2_W0], H, Gr[CAMO_P_F2_B0], 2 * H, H, TK);
    g.tn(f.dU2_16, H, f.O2_16, H, Gr[CAMO_P_A2_OUT_W], H, Gr[CAMO_P_A2_OUT_B], H, H, TK);
    g.tn(f.dQKVkg16, 3 * H, f.G16, H, Gr[CAMO_P_A2_IN_W], H, Gr[CAMO_P_A2_IN_B], H, H, TK);
    g.tn(f.dQKVkg16 + H, 3 * H, f.G16, H, Gr[CAMO_P_A1_IN_W] + HH, H, Gr[CAMO_P_A1_IN_B] + H, 2 * H, H, TK);
    g.tn(f.dG16, H, f.KG16, D, Gr[CAMO_P_KG_PROJ_W], D, Gr[CAMO_P_KG_PROJ_B], H, D, TK);
    CK(g.run(), "node-level weight gradients");
    return 0;
  }
  // Node-level weight gradients: dW += dy^T . x over the rows of a stream (bf16 operands the fused kernels wrote).  The input
  // projection and the in-projections take theirs in PARAMETER space -- with R = x W_p^T + b_p and [q|k'|v'] = R W_in^T + b_in:
  //   M = dQKV^T x  [3H][D],  db_in = colsum(dQKV);   dW_in = dQKV^T R = M W_p^T + db_in b_p^T;
  //   dW_p = dR^T x = dU^T x + W_in^T M,  db_p = colsum(dU) + W_in^T db_in      (dR = dU + dQKV W_in never exists)
  // -- 131 k MACs per row (M, dU^T x) instead of 427 k (dR, dQKV^T R, dR^T x), and one small fp32 launch behind them (misc.hip, unfold_kernel).
  float* const Mrg = w.parM; float* const Mkg = Mrg + (size_t)3 * H * D;
  float* const dbrg = Mkg + (size_t)3 * H * D; float* const dbkg = dbrg + 3 * H;
  GB16 g(drop, st);
  g.tn(f.dH16, 2 * H, f.Y16, H, Gr[CAMO_P_F1_W0], H, Gr[CAMO_P_F1_B0], 2 * H, H, T);
  g.tn(f.dU16, H, f.O16, H, Gr[CAMO_P_A1_OUT_W], H, Gr[CAMO_P_A1_OUT_B], H, H, T);
  g.tn(f.dQKV16, 3 * H, f.X16, D, Mrg, D, dbrg, H, D, T).bias_grad2 = Gr[CAMO_P_A1_IN_B];
  g.tn(f.dQKV16 + H, 3 * H, f.X16, D, Mrg + (size_t)H * D, D, dbrg + H, 2 * H, D, T).bias_grad2 = Gr[CAMO_P_A2_IN_B] + H;
  g.tn(f.dU16, H, f.X16, D, Gr[CAMO_P_RG_PROJ_W], D, Gr[CAMO_P_RG_PROJ_B], H, D, T);
  g.tn(f.dH2_16, 2 * H, f.Y2_16, H, Gr[CAMO_P_F2_W0], H, Gr[CAMO_P_F2_B0], 2 * H, H, TK);
  g.tn(f.dU2_16, H, f.O2_16, H, Gr[CAMO_P_A2_OUT_W], H, Gr[CAMO_P_A2_OUT_B], H, H, TK);
  g.tn(f.dQKVkg16, 3 * H, f.KG16, D, Mkg, D, dbkg, H, D, TK).bias_grad2 = Gr[CAMO_P_A2_IN_B];
  g.tn(f.dQKVkg16 + H, 3 * H, f.KG16, D, Mkg + (size_t)H * D, D, dbkg + H, 2 * H, D, TK).bias_grad2 = Gr[CAMO_P_A1_IN_B] + H;
  g.tn(f.dU2_16, H, f.KG16, D, Gr[CAMO_P_KG_PROJ_W], D, Gr[CAMO_P_KG_PROJ_B], H, D, TK);
  CK(g.run(), "node-level weight gradients");
  {
    const UnfoldStream urg{Mrg, dbrg, P[CAMO_P_A1_IN_W], P[CAMO_P_A2_IN_W] + HH, P[CAMO_P_RG_PROJ_W], P[CAMO_P_RG_PROJ_B], Gr[CAMO_P_A1_IN_W],
                           Gr[CAMO_P_A2_IN_W] + HH, Gr[CAMO_P_RG_PROJ_W], Gr[CAMO_P_RG_PROJ_B]};
    const UnfoldStream ukg{Mkg, dbkg, P[CAMO_P_A2_IN_W], P[CAMO_P_A1_IN_W] + HH, P[CAMO_P_KG_PROJ_W], P[CAMO_P_KG_PROJ_B], Gr[CAMO_P_A2_IN_W],
                           Gr[CAMO_P_A1_IN_W] + HH, Gr[CAMO_P_KG_PROJ_W], Gr[CAMO_P_KG_PROJ_B]};
    CK(launch_unfold(urg, ukg, st), "projection / in-projection weight gradients (parameter space)");
  }
  return 0;
}

}  // namespace

extern "C" {

int camo_abi_version(void) { return CAMO_ABI_VERSION; }
const char* camo_last_error(void) { return g_err.c_str(); }

size_t camo_workspace_bytes(const camo_dims_t* dims, int32_t B, int32_t T, int32_t Nk) {
  OptScope opt_scope(dims);
  if (check_dims(dims, B, T, Nk)) return 0;
  return carve(*dims, B, T, Nk, nullptr).bytes;
}

size_t camo_batch_desc_bytes(int32_t B, int32_t T) {
  if (B < 1 || T < B) { fail(CAMO_E_ARG, "need B >= 1 and T >= B"); return 0; }
  return desc_carve(B, T, nullptr).bytes;
}

int camo_prepare_batch(const int32_t* rg_offsets, int32_t B, int32_t T, int32_t max_nr, void* desc, size_t desc_bytes, void* stream) {
  if (!rg_offsets || !desc || B < 1 || T < B || max_nr < 1 || max_nr > T) return fail(CAMO_E_ARG, "bad prepare_batch arguments");
  const Desc d = desc_carve(B, T, desc);
  if (desc_bytes < d.bytes) return fail(CAMO_E_WORKSPACE, "descriptor buffer smaller than camo_batch_desc_bytes()");
  CK(launch_rowmap(rg_offsets, d.row_sample, d.inv_nr, d.tile_off, d.tile_desc, B, T / 32 + B, max_nr, static_cast<hipStream_t>(stream)), "rowmap");
  return 0;
}

int camo_gather_batch(const float* rg_all, const int64_t* sample_offsets, const float* kg_all, const int64_t* y_all, const float* e_all, const float* s_all,
                      const int64_t* idx, int32_t B, int32_t T, int32_t rg_dim, int32_t kg_floats, float* rg_out, float* kg_out, int32_t* offsets_out,
                      int64_t* y_out, float* e_out, float* s_out, float noise_std, uint64_t seed, void* stream) {
  if (!rg_all || !sample_offsets || !kg_all || !y_all || !e_all || !s_all || !idx || !rg_out || !kg_out || !offsets_out || !y_out || !e_out || !s_out)
    return fail(CAMO_E_ARG, "null pointer argument");
  if (B < 1 || T < B) return fail(CAMO_E_ARG, "need B >= 1 and T >= B");
  if (B > 4096 || rg_dim < 4 || (rg_dim & 3) || rg_dim > 1024 || kg_floats < 2 || (kg_floats & 1) || noise_std < 0.f)
    return fail(CAMO_E_UNSUPPORTED, "camo_gather_batch: B <= 4096, rg_dim a multiple of 4 (<= 1024), an even number of KG floats per sample");
  CK(launch_gather_batch(rg_all, reinterpret_cast<const long long*>(sample_offsets), kg_all, reinterpret_cast<const long long*>(y_all), e_all, s_all,
                         reinterpret_cast<const long long*>(idx), B, T, rg_dim, kg_floats, rg_out, kg_out, offsets_out, reinterpret_cast<long long*>(y_out),
                         e_out, s_out, noise_std, seed, static_cast<hipStream_t>(stream)), "gather batch");
  return 0;
}

static int forward_impl(const camo_dims_t* dims, const float* const* params, const float* rg, const int32_t* rg_offsets,
                        const void* desc,
                        const float* kg, int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace,
                        size_t workspace_bytes, float* outs, float* attn_rg2kg, float* attn_kg2rg, int32_t training,
                        uint64_t seed, int32_t precision, int32_t flags, void* stream, const FusedLoss* fl,
                        const FusedLoss* fl17 = nullptr /* given: loss + the whole tail backward ride in the one-launch tail */) {
  t_tailw_bwd_planes = false;
  if (int e = check_dims(dims, B, T, Nk)) return e;
  if (!params || !rg || !rg_offsets || !desc || !kg || !workspace || !outs)
    return fail(CAMO_E_ARG, "null pointer argument");
  const Desc bd = desc_carve(B, T, const_cast<void*>(desc));
  const int32_t* row_sample = bd.row_sample; const float* inv_nr = bd.inv_nr;
  if (max_nr < 1 || max_nr > T) return fail(CAMO_E_ARG, "max_nr out of range");
  if (precision != CAMO_PREC_F32 && precision != CAMO_PREC_BF16) return fail(CAMO_E_ARG, "unknown precision");
  const camo_dims_t& d = *dims;
  Ws w = carve(d, B, T, Nk, workspace);
  bind_shadows(w);
  if (workspace_bytes < w.bytes) return fail(CAMO_E_WORKSPACE, "workspace smaller than camo_workspace_bytes()");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const DropCfg drop = make_drop(training, d.dropout, seed);
  const int H = d.hidden_dim, D = d.rg_dim, Dk = d.kg_dim, TK = B * Nk, nh = d.num_heads;
  const float* const* P = params;
  GB g(drop, precision, st);
  GB gt(drop, CAMO_PREC_F32, st);   // per-sample (B-row) GEMMs
  // one clear of everything this step accumulates into with atomics (means now, dfused/dKV in backward);
  // the bf16 schedule's prep launch does it along with its casts
  // calls at the reference configuration that do not ask for attention maps take the fused row-tile schedule
  const bool use17 = !attn_rg2kg && !attn_kg2rg && !(flags & CAMO_FLAG_ATTN_MAPS) && fused17_ok(d, P, precision, Nk, max_nr);
  const bool save17 = !(flags & CAMO_FWD_INFERENCE) || g_opt_fused_save != 0;
  const bool use16 = !use17 && sched16_ok(d, P, precision, T, Nk, max_nr);
  if (!use16 && !use17) CK((int)hipMemsetAsync(w.zero_base, 0, w.zero_bytes, st), "memset zero block");

  if (d.fusion_type == CAMO_FUSION_LATE) {
    // LateFusion.forward, fusion_model.py:164-171
    const int F = H / 2, Dc = D + Dk;
    SegMean sm[2] = {{rg, D, D, rg_offsets, 0, w.comb, Dc}, {kg, Dk, Dk, nullptr, Nk, w.comb + D, Dc}};
    CK(launch_seg_mean(sm, 2, B, max_nr > Nk ? max_nr : Nk, st), "late means");
    set_drop(gt.nt(w.comb, Dc, P[CAMO_PL_W0], Dc, P[CAMO_PL_B0], w.F1, H, B, H, Dc, GF_RELU), SITE_LATE0);
    CK(gt.run(), "late fc0");
    set_drop(gt.nt(w.F1, H, P[CAMO_PL_W3], H, P[CAMO_PL_B3], w.a2, F, B, F, H, GF_RELU), SITE_LATE0 + 1);
    CK(gt.run(), "late fc3");
    gt.nt(w.a2, F, P[CAMO_PL_W6], F, P[CAMO_PL_B6], w.fused, F, B, F, F);
    CK(gt.run(), "late fc6");
    return heads_forward(d, P + CAMO_PL_HEADS, w, B, F, outs, gt, fl);
  }

  // ---- CrossAttentionFusion.forward, fusion_model.py:75-146
  const float* R = rg; const float* G = kg;
  if (!P[CAMO_P_RG_PROJ_W] && D != H) return fail(CAMO_E_ARG, "rg_proj weight missing but rg_dim != hidden_dim");
  if (!P[CAMO_P_KG_PROJ_W] && Dk != H) return fail(CAMO_E_ARG, "kg_proj weight missing but kg_dim != hidden_dim");
  const size_t HH2 = (size_t)H * H;
  if (use17) {
    const bool tailw = (flags & CAMO_FWD_INFERENCE) && !fl && !fl17 && tailw_taken(d, B, T, max_nr, save17, drop.p > 0.f);
    // training calls on the wide front half with more than 64 samples: the tail's FORWARD as the one two-plane launch (with fp32
    // copies of what the backward launches read) instead of four fp32 GEMM launches (B = 256: 4 x 27 us -> 34 us); the loss launch
    // and the backward launches follow as before
    // (its weight planes are built by extra blocks of the wide front launch: the same predicate as forward_nodes17's)
    const bool tailw_train = !tailw && fl && !fl17 && save17 && B > 64 && g_opt_tailw != 0 && T >= 4 * 32 * 224 && wide_train_front_rt(T, max_nr, save17) != 0 &&
                             tail_wide_ok(B, d.num_classes) && heads_loss_ok(B, d.num_classes);
    if (int e = forward_nodes17(d, P, rg, rg_offsets, bd, kg, B, T, Nk, max_nr, w, drop, save17, st, tailw_train ? 2 : (tailw ? 1 : 0))) return e;
    t_tailw_bwd_planes = tailw_train;        // (the backward half of this training call may take the two-plane launch too)
    if (tailw || tailw_train) {
      TailWideArgs ta; std::memset(&ta, 0, sizeof(ta));
      ta.Ymean = w.Ymean; ta.H1mean = w.H1mean; ta.Y2mean = w.Y2mean; ta.H2mean = w.H2mean;
      const us* const* tw = w.f.tailw;
      ta.T13h = tw[0]; ta.T13l = tw[1]; ta.T23h = tw[2]; ta.T23l = tw[3]; ta.Tfu0h = tw[4]; ta.Tfu0l = tw[5]; ta.Tfu3h = tw[6]; ta.Tfu3l = tw[7]; ta.Th0h = tw[8]; ta.Th0l = tw[9];
      ta.b13 = P[CAMO_P_F1_B3]; ta.b23 = P[CAMO_P_F2_B3]; ta.bfu0 = P[CAMO_P_FU_B0]; ta.bfu3 = P[CAMO_P_FU_B3];
      for (int x = 0; x < 4; ++x) { ta.bh0[x] = P[CAMO_P_HEADS + 4 * x + 1]; ta.Wh3[x] = P[CAMO_P_HEADS + 4 * x + 2]; ta.bh3[x] = P[CAMO_P_HEADS + 4 * x + 3]; }
      ta.outs = outs; ta.B = B; ta.C = d.num_classes; ta.drop = drop;
      if (tailw_train) { ta.comb_out = w.comb; ta.F1_out = w.F1; ta.fused_out = w.fused; ta.hid_out = w.hid; }
      CK(launch_tail_wide(ta, st), "per-sample tail (wide, one launch)");
      if (!tailw_train) return 0;
      // (the output layers' weight gradients ride in the tail's weight-gradient launch of the backward half, when that half takes it)
      return heads_forward(d, P + CAMO_P_HEADS, w, B, H, outs, gt, fl, /*hidden_done=*/true, /*defer_out_grads=*/g_opt_tailw_bwd != 0);
    }
  } else if (use16) {
    if (int e = forward_nodes16(d, P, rg, rg_offsets, row_sample, inv_nr, kg, B, T, Nk, max_nr, w, attn_rg2kg, attn_kg2rg, drop, st)) return e;
  } else {
  if (P[CAMO_P_KG_PROJ_W]) { g.nt(kg, Dk, P[CAMO_P_KG_PROJ_W], Dk, P[CAMO_P_KG_PROJ_B], w.G, H, TK, H, Dk); G = w.G; }
  if (P[CAMO_P_RG_PROJ_W]) { g.nt(rg, D, P[CAMO_P_RG_PROJ_W], D, P[CAMO_P_RG_PROJ_B], w.R, H, T, H, D); R = w.R; }
  CK(g.run(), "input projections");
  // in-projections of both attention blocks (packed in_proj_weight: rows 0..H-1 = Wq, H..3H-1 = Wk|Wv)
  g.nt(R, H, P[CAMO_P_A1_IN_W], H, P[CAMO_P_A1_IN_B], w.Q, H, T, H, H);
  g.nt(R, H, P[CAMO_P_A2_IN_W] + HH2, H, P[CAMO_P_A2_IN_B] + H, w.KV2, 2 * H, T, 2 * H, H);
  g.nt(G, H, P[CAMO_P_A1_IN_W] + HH2, H, P[CAMO_P_A1_IN_B] + H, w.KV, 2 * H, TK, 2 * H, H);
  g.nt(G, H, P[CAMO_P_A2_IN_W], H, P[CAMO_P_A2_IN_B], w.Q2, H, TK, H, H);
  CK(g.run(), "attention in-projections");
      CK(launch_attn_rg2kg_fwd(w.Q, w.KV, rg_offsets, w.P, w.O, attn_rg2kg, B, T, max_nr, H, nh, Nk, drop, st), "attn rg2kg fwd");
  CK(launch_attn_kg2rg_fwd(w.Q2, w.KV2, rg_offsets, w.P2, w.O2, B, max_nr, H, nh, Nk, drop, st), "attn kg2rg fwd");
  if (attn_kg2rg) CK(launch_attn_avg(w.P2, attn_kg2rg, T, nh, Nk, drop, st), "attn avg");
  // out-projection + residual (fusion_model.py:119,130), then LayerNorm
  set_res(g.nt(w.O, H, P[CAMO_P_A1_OUT_W], H, P[CAMO_P_A1_OUT_B], w.U, H, T, H, H), R, H);
  set_res(g.nt(w.O2, H, P[CAMO_P_A2_OUT_W], H, P[CAMO_P_A2_OUT_B], w.U2, H, TK, H, H), G, H);
  CK(g.run(), "attention out-projections");
  {
    LnSeg s0{w.U, w.Y, w.st1, P[CAMO_P_LN1_W], P[CAMO_P_LN1_B], T};
    LnSeg s1{w.U2, w.Y2, w.st2, P[CAMO_P_LN2_W], P[CAMO_P_LN2_B], TK};
    CK(launch_ln_fwd(s0, s1, H, st), "layernorm fwd");
  }
  // FFN first layers (ReLU + dropout fused), fusion_model.py:53-65
  set_drop(g.nt(w.Y, H, P[CAMO_P_F1_W0], H, P[CAMO_P_F1_B0], w.H1, 2 * H, T, 2 * H, H, GF_RELU), SITE_FFN_RG);
  set_drop(g.nt(w.Y2, H, P[CAMO_P_F2_W0], H, P[CAMO_P_F2_B0], w.H2, 2 * H, TK, 2 * H, H, GF_RELU), SITE_FFN_KG);
  CK(g.run(), "ffn layer 0");
  }
  // per-sample means of Y and H1d, then the second FFN layer on the means (mean-pool linearity)
  if (use17 && g_opt_tail17 != 0 && (fl17 || ((flags & CAMO_FWD_INFERENCE) && !fl)) && tail_fused_ok(B, d.num_classes))
    return tail17(d, P, fl17 ? fl17->head_grads - CAMO_P_HEADS : nullptr, w, B, outs, fl17, drop, st);
  if (use16 || use17) {
    // accumulated by the kernels that produce the pooled tensors
  } else {
    SegMean sm[4] = {{w.Y, H, H, rg_offsets, 0, w.Ymean, H}, {w.H1, 2 * H, 2 * H, rg_offsets, 0, w.H1mean, 2 * H},
                     {w.Y2, H, H, nullptr, Nk, w.Y2mean, H}, {w.H2, 2 * H, 2 * H, nullptr, Nk, w.H2mean, 2 * H}};
    CK(launch_seg_mean(sm, 4, B, max_nr > Nk ? max_nr : Nk, st), "pool");
  }
  set_res(gt.nt(w.H1mean, 2 * H, P[CAMO_P_F1_W3], 2 * H, P[CAMO_P_F1_B3], w.comb, 2 * H, B, H, 2 * H), w.Ymean, H);
  set_res(gt.nt(w.H2mean, 2 * H, P[CAMO_P_F2_W3], 2 * H, P[CAMO_P_F2_B3], w.comb + H, 2 * H, B, H, 2 * H), w.Y2mean, H);
  CK(gt.run(), "ffn layer 3 on pooled rows");
  // fusion layer (fusion_model.py:68-73,138-139)
  set_drop(gt.nt(w.comb, 2 * H, P[CAMO_P_FU_W0], 2 * H, P[CAMO_P_FU_B0], w.F1, H, B, H, 2 * H, GF_RELU), SITE_FUSE);
  CK(gt.run(), "fusion layer 0");
  gt.nt(w.F1, H, P[CAMO_P_FU_W3], H, P[CAMO_P_FU_B3], w.fused, H, B, H, H);
  CK(gt.run(), "fusion layer 3");
  return heads_forward(d, P + CAMO_P_HEADS, w, B, H, outs, gt, fl);
}

int camo_forward(const camo_dims_t* dims, const float* const* params, const float* rg, const int32_t* rg_offsets,
                 const void* batch_desc,
                 const float* kg, int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace,
                 size_t workspace_bytes, float* outs, float* attn_rg2kg, float* attn_kg2rg, int32_t training,
                 uint64_t seed, int32_t precision, int32_t flags, void* stream) {
  OptScope opt_scope(dims);
  return forward_impl(dims, params, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes, outs,
                      attn_rg2kg, attn_kg2rg, training, seed, precision, flags, stream, nullptr);
}

int camo_forward_cached(const camo_dims_t* dims, const float* const* params, const float* rg, const int32_t* rg_offsets,
                        const void* batch_desc, const float* kg, int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace,
                        size_t workspace_bytes, float* outs, float* attn_rg2kg, float* attn_kg2rg, int32_t training,
                        uint64_t seed, int32_t precision, int32_t flags, void* shadows, int32_t shadows_valid, int32_t* shadows_state,
                        void* stream) {
  OptScope opt_scope(dims);
  if (shadows_state) *shadows_state = 0;
  if (shadows_valid && !shadows) return fail(CAMO_E_ARG, "shadows_valid without a shadow buffer");
  // A call that saves for camo_backward would leave the backward's transposed shadows in the caller's buffer, where camo_backward
  // (which takes no shadow argument) cannot find them, and its deferred clears in thread-local state that any other forward call
  // drops: the training pair is camo_forward_loss_backward, which owns both halves.
  if (shadows && !(flags & CAMO_FWD_INFERENCE))
    return fail(CAMO_E_UNSUPPORTED, "camo_forward_cached with a shadow buffer serves inference calls only (flags must contain CAMO_FWD_INFERENCE)");
  if (shadows && (reinterpret_cast<uintptr_t>(shadows) & 255)) return fail(CAMO_E_ARG, "the shadow buffer must be 256-byte aligned");
  t_shadows = shadows; t_shadows_valid = shadows && shadows_valid != 0; t_fold_missing = shadows && shadows_valid == 2; t_shadows_state = 0;
  const int rc = forward_impl(dims, params, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes, outs,
                              attn_rg2kg, attn_kg2rg, training, seed, precision, flags, stream, nullptr);
  t_shadows = nullptr; t_shadows_valid = false; t_fold_missing = false;
  if (rc == 0 && shadows_state) *shadows_state = t_shadows_state;
  return rc;
}

// camo_forward_loss_backward's optional event: recorded on the stream as soon as the gradients of the per-sample tail (pooled
// FFN layers, fusion layer, heads: parameters CAMO_P_F2_W3 .. end of the table, and CAMO_P_F1_W3/B3) are final, so that a
// data-parallel caller can start reducing that part of the flat buffer while the node-level backward runs.
static thread_local hipEvent_t t_tail_event = nullptr;
static int record_tail_event(hipStream_t st) {
  if (!t_tail_event) return 0;
  const hipEvent_t ev = t_tail_event; t_tail_event = nullptr;
  return (int)hipEventRecord(ev, st);
}

static int backward_impl(const camo_dims_t* dims, const float* const* params, float* const* grads, const float* rg,
                         const int32_t* rg_offsets, const void* desc, const float* kg, int32_t B,
                         int32_t T, int32_t Nk, int32_t max_nr, void* workspace, size_t workspace_bytes, const float* outs,
                         const float* d_outs, int32_t d_outs_pre_activation, int32_t training, uint64_t seed, int32_t precision,
                         int32_t flags, void* stream, bool heads_out_done) {
  if (int e = check_dims(dims, B, T, Nk)) return e;
  if (!params || !grads || !rg || !rg_offsets || !desc || !kg || !workspace || !outs || (!d_outs && !heads_out_done))
    return fail(CAMO_E_ARG, "null pointer argument");
  const Desc bd = desc_carve(B, T, const_cast<void*>(desc));
  const int32_t* row_sample = bd.row_sample; const float* inv_nr = bd.inv_nr;
  if (max_nr < 1 || max_nr > T) return fail(CAMO_E_ARG, "max_nr out of range");
  if (precision != CAMO_PREC_F32 && precision != CAMO_PREC_BF16) return fail(CAMO_E_ARG, "unknown precision");
  const camo_dims_t& d = *dims;
  Ws w = carve(d, B, T, Nk, workspace);
  bind_shadows(w);
  if (workspace_bytes < w.bytes) return fail(CAMO_E_WORKSPACE, "workspace smaller than camo_workspace_bytes()");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const DropCfg drop = make_drop(training, d.dropout, seed);
  const int H = d.hidden_dim, D = d.rg_dim, Dk = d.kg_dim, TK = B * Nk, nh = d.num_heads;
  const float* const* P = params;
  float* const* Gr = grads;
  GB g(drop, precision, st);
  GB gt(drop, CAMO_PREC_F32, st);   // per-sample (B-row) GEMMs

  if (d.fusion_type == CAMO_FUSION_LATE) {
    const int F = H / 2, Dc = D + Dk;
    if (int e = heads_backward(d, P + CAMO_PL_HEADS, Gr + CAMO_PL_HEADS, w, B, F, outs, d_outs, d_outs_pre_activation, drop, st, gt, heads_out_done)) return e;
    set_relu_bwd(gt.nn(w.dfused, F, P[CAMO_PL_W6], F, w.da2, F, B, F, F), w.a2, F, drop.scale);
    gt.tn(w.dfused, F, w.a2, F, Gr[CAMO_PL_W6], F, Gr[CAMO_PL_B6], F, F, B);
    CK(gt.run(), "late fc6 bwd");
    set_relu_bwd(gt.nn(w.da2, F, P[CAMO_PL_W3], H, w.dF1, H, B, H, F), w.F1, H, drop.scale);
    gt.tn(w.da2, F, w.F1, H, Gr[CAMO_PL_W3], H, Gr[CAMO_PL_B3], F, H, B);
    CK(gt.run(), "late fc3 bwd");
    gt.tn(w.dF1, H, w.comb, Dc, Gr[CAMO_PL_W0], Dc, Gr[CAMO_PL_B0], H, Dc, B);
    CK(gt.run(), "late fc0 bwd");
    return 0;
  }

  const bool has_rgp = P[CAMO_P_RG_PROJ_W] != nullptr, has_kgp = P[CAMO_P_KG_PROJ_W] != nullptr;
  const float* R = has_rgp ? w.R : rg;
  const float* G = has_kgp ? w.G : kg;
  const bool tailw_bwd = t_tailw_bwd_planes && heads_out_done && g_opt_tailw_bwd != 0;
  t_tailw_bwd_planes = false;
  if (tailw_bwd) {
    // the tail's input-gradient chain as ONE two-plane launch (tail_wide.h) + ONE launch for its weight gradients, instead of four
    // fp32 GEMM launches that each pair an input gradient with a weight gradient (B = 256: 108 us)
    TailWideBwdArgs ta; std::memset(&ta, 0, sizeof(ta));
    const us* const* tw = w.f.tailw;
    ta.dhid = w.dhid; ta.F1 = w.F1;
    ta.Th0h = tw[10]; ta.Th0l = tw[11]; ta.Tfu3h = tw[12]; ta.Tfu3l = tw[13]; ta.Tfu0h = tw[14]; ta.Tfu0l = tw[15];
    ta.T13h = tw[16]; ta.T13l = tw[17]; ta.T23h = tw[18]; ta.T23l = tw[19];
    ta.dfused = w.dfused; ta.dF1 = w.dF1; ta.dcomb = w.dcomb; ta.dHm1 = w.dHm1; ta.dHm2 = w.dHm2;
    ta.B = B; ta.scale = drop.scale;
    CK(launch_tail_wide_bwd(ta, st), "per-sample tail, input gradients (wide, one launch)");
    const int Fh = H / 2;
    float* const* hg = Gr + CAMO_P_HEADS;
    {
      const int C = d.num_classes, Wd = 2 * C + 2, nout[4] = {C, C, 1, 1}, coff[4] = {0, C, 2 * C, 2 * C + 1};
      for (int x = 0; x < 4; ++x) gt.tn(w.dlog + coff[x], Wd, w.hid + x * Fh, 4 * Fh, hg[4 * x + 2], Fh, hg[4 * x + 3], nout[x], Fh, B);   // (left by the loss launch)
    }
    for (int x = 0; x < 4; ++x) gt.tn(w.dhid + x * Fh, 4 * Fh, w.fused, H, hg[4 * x], H, hg[4 * x + 1], Fh, H, B);
    gt.tn(w.dfused, H, w.F1, H, Gr[CAMO_P_FU_W3], H, Gr[CAMO_P_FU_B3], H, H, B);
    gt.tn(w.dF1, H, w.comb, 2 * H, Gr[CAMO_P_FU_W0], 2 * H, Gr[CAMO_P_FU_B0], H, 2 * H, B);
    gt.tn(w.dcomb, 2 * H, w.H1mean, 2 * H, Gr[CAMO_P_F1_W3], 2 * H, Gr[CAMO_P_F1_B3], H, 2 * H, B);
    gt.tn(w.dcomb + H, 2 * H, w.H2mean, 2 * H, Gr[CAMO_P_F2_W3], 2 * H, Gr[CAMO_P_F2_B3], H, 2 * H, B);
    CK(gt.run(), "per-sample tail, weight gradients");
  } else {
  if (int e = heads_backward(d, P + CAMO_P_HEADS, Gr + CAMO_P_HEADS, w, B, H, outs, d_outs, d_outs_pre_activation, drop, st, gt, heads_out_done)) return e;
  // fusion layer
  set_relu_bwd(gt.nn(w.dfused, H, P[CAMO_P_FU_W3], H, w.dF1, H, B, H, H), w.F1, H, drop.scale);
  gt.tn(w.dfused, H, w.F1, H, Gr[CAMO_P_FU_W3], H, Gr[CAMO_P_FU_B3], H, H, B);
  CK(gt.run(), "fusion layer 3 bwd");
  gt.nn(w.dF1, H, P[CAMO_P_FU_W0], 2 * H, w.dcomb, 2 * H, B, 2 * H, H);
  gt.tn(w.dF1, H, w.comb, 2 * H, Gr[CAMO_P_FU_W0], 2 * H, Gr[CAMO_P_FU_B0], H, 2 * H, B);
  CK(gt.run(), "fusion layer 0 bwd");
  // pooled second FFN layer: d(mean H1d) = dpool.W2 ; dW2 += dpool^T.mean(H1d) ; db2 += sum_b dpool
  gt.nn(w.dcomb, 2 * H, P[CAMO_P_F1_W3], 2 * H, w.dHm1, 2 * H, B, 2 * H, H);
  gt.nn(w.dcomb + H, 2 * H, P[CAMO_P_F2_W3], 2 * H, w.dHm2, 2 * H, B, 2 * H, H);
  gt.tn(w.dcomb, 2 * H, w.H1mean, 2 * H, Gr[CAMO_P_F1_W3], 2 * H, Gr[CAMO_P_F1_B3], H, 2 * H, B);
  gt.tn(w.dcomb + H, 2 * H, w.H2mean, 2 * H, Gr[CAMO_P_F2_W3], 2 * H, Gr[CAMO_P_F2_B3], H, 2 * H, B);
  CK(gt.run(), "ffn layer 3 bwd (pooled)");
  }
  CK(record_tail_event(st), "tail event");
  if (!(flags & CAMO_FLAG_ATTN_MAPS) && fused17_ok(d, P, precision, Nk, max_nr))
    return backward_nodes17(d, P, Gr, rg_offsets, bd, B, T, Nk, w, drop, st);
  if (sched16_ok(d, P, precision, T, Nk, max_nr))
    return backward_nodes16(d, P, Gr, rg_offsets, row_sample, inv_nr, B, T, Nk, max_nr, w, drop, st);
  {
    BcastSeg s0{w.H1, w.dHm1, 2 * H, row_sample, inv_nr, 0, w.dH1, T};
    BcastSeg s1{w.H2, w.dHm2, 2 * H, nullptr, nullptr, Nk, w.dH2, TK};
    CK(launch_relu_bcast_bwd(s0, s1, 2 * H, drop.scale, st), "relu bcast bwd");
  }
  // first FFN layer: dY = bcast(dpool)/n + dH1.W1 ; dW1 += dH1^T.Y
  set_bcast(g.nn(w.dH1, 2 * H, P[CAMO_P_F1_W0], H, w.dY, H, T, H, 2 * H), w.dcomb, 2 * H, row_sample, inv_nr, 0);
  set_bcast(g.nn(w.dH2, 2 * H, P[CAMO_P_F2_W0], H, w.dY2, H, TK, H, 2 * H), w.dcomb + H, 2 * H, nullptr, nullptr, Nk);
  g.tn(w.dH1, 2 * H, w.Y, H, Gr[CAMO_P_F1_W0], H, Gr[CAMO_P_F1_B0], 2 * H, H, T);
  g.tn(w.dH2, 2 * H, w.Y2, H, Gr[CAMO_P_F2_W0], H, Gr[CAMO_P_F2_B0], 2 * H, H, TK);
  CK(g.run(), "ffn layer 0 bwd");
  {
    LnBwdSeg s0{w.U, w.dY, w.st1, P[CAMO_P_LN1_W], w.dU, Gr[CAMO_P_LN1_W], Gr[CAMO_P_LN1_B], T};
    LnBwdSeg s1{w.U2, w.dY2, w.st2, P[CAMO_P_LN2_W], w.dU2, Gr[CAMO_P_LN2_W], Gr[CAMO_P_LN2_B], TK};
    CK(launch_ln_bwd(s0, s1, H, st), "layernorm bwd");
  }
  // out-projections
  g.nn(w.dU, H, P[CAMO_P_A1_OUT_W], H, w.dO, H, T, H, H);
  g.nn(w.dU2, H, P[CAMO_P_A2_OUT_W], H, w.dO2, H, TK, H, H);
  g.tn(w.dU, H, w.O, H, Gr[CAMO_P_A1_OUT_W], H, Gr[CAMO_P_A1_OUT_B], H, H, T);
  g.tn(w.dU2, H, w.O2, H, Gr[CAMO_P_A2_OUT_W], H, Gr[CAMO_P_A2_OUT_B], H, H, TK);
  CK(g.run(), "out-projection bwd");
  CK(launch_attn_rg2kg_bwd(w.Q, w.KV, w.P, w.dO, rg_offsets, w.dQ, w.dKV, B, max_nr, H, nh, Nk, drop, st), "attn rg2kg bwd");
  CK(launch_attn_kg2rg_bwd(w.Q2, w.KV2, w.P2, w.dO2, w.O2, rg_offsets, w.dQ2, w.dKV2, w.dS2, B, max_nr, H, nh, Nk, drop, st), "attn kg2rg bwd");
  // in-projection weight gradients, and the gradients flowing into R and G
  const size_t HH = (size_t)H * H;
  g.tn(w.dQ, H, R, H, Gr[CAMO_P_A1_IN_W], H, Gr[CAMO_P_A1_IN_B], H, H, T);
  g.tn(w.dKV2, 2 * H, R, H, Gr[CAMO_P_A2_IN_W] + HH, H, Gr[CAMO_P_A2_IN_B] + H, 2 * H, H, T);
  g.tn(w.dKV, 2 * H, G, H, Gr[CAMO_P_A1_IN_W] + HH, H, Gr[CAMO_P_A1_IN_B] + H, 2 * H, H, TK);
  g.tn(w.dQ2, H, G, H, Gr[CAMO_P_A2_IN_W], H, Gr[CAMO_P_A2_IN_B], H, H, TK);
  if (has_rgp) set_res(g.nn(w.dQ, H, P[CAMO_P_A1_IN_W], H, w.dR, H, T, H, H), w.dU, H);
  if (has_kgp) set_res(g.nn(w.dQ2, H, P[CAMO_P_A2_IN_W], H, w.dG, H, TK, H, H), w.dU2, H);
  CK(g.run(), "in-projection bwd");
  if (has_rgp) set_res(g.nn(w.dKV2, 2 * H, P[CAMO_P_A2_IN_W] + HH, H, w.dR, H, T, H, 2 * H), w.dR, H);
  if (has_kgp) set_res(g.nn(w.dKV, 2 * H, P[CAMO_P_A1_IN_W] + HH, H, w.dG, H, TK, H, 2 * H), w.dG, H);
  CK(g.run(), "k|v input gradients");
  if (has_rgp) g.tn(w.dR, H, rg, D, Gr[CAMO_P_RG_PROJ_W], D, Gr[CAMO_P_RG_PROJ_B], H, D, T);
  if (has_kgp) g.tn(w.dG, H, kg, Dk, Gr[CAMO_P_KG_PROJ_W], Dk, Gr[CAMO_P_KG_PROJ_B], H, Dk, TK);
  CK(g.run(), "input projection bwd");
  return 0;
}

int camo_backward(const camo_dims_t* dims, const float* const* params, float* const* grads, const float* rg,
                  const int32_t* rg_offsets, const void* batch_desc, const float* kg, int32_t B,
                  int32_t T, int32_t Nk, int32_t max_nr, void* workspace, size_t workspace_bytes, const float* outs,
                  const float* d_outs, int32_t d_outs_pre_activation, int32_t training, uint64_t seed, int32_t precision,
                  int32_t flags, void* stream) {
  OptScope opt_scope(dims);
  return backward_impl(dims, params, grads, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes, outs,
                       d_outs, d_outs_pre_activation, training, seed, precision, flags, stream, false);
}

static int forward_loss_backward_impl(const camo_dims_t* dims, const float* const* params, float* const* grads, const float* rg,
                               const int32_t* rg_offsets, const void* batch_desc, const float* kg,
                               int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace, size_t workspace_bytes,
                               const int64_t* y, const float* e, const float* s, float* outs, float* loss_terms, int32_t* pred,
                               int32_t training, uint64_t seed, int32_t precision, void* stream);

int camo_forward_loss_backward(const camo_dims_t* dims, const float* const* params, float* const* grads, const float* rg,
                               const int32_t* rg_offsets, const void* batch_desc, const float* kg,
                               int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace, size_t workspace_bytes,
                               const int64_t* y, const float* e, const float* s, float* outs, float* loss_terms, int32_t* pred,
                               int32_t training, uint64_t seed, int32_t precision, void* tail_event, void* shadows,
                               int32_t shadows_valid, void* stream) {
  OptScope opt_scope(dims);
  if (!dims || !grads || !y || !e || !s || !outs || !loss_terms) return fail(CAMO_E_ARG, "null pointer argument");
  if (shadows_valid && !shadows) return fail(CAMO_E_ARG, "shadows_valid without a shadow buffer");
  // external shadows are used by the fused schedule only; whether the call takes it is known from its arguments
  const bool ext = shadows && params && !check_dims(dims, B, T, Nk) && fused17_ok(*dims, params, precision, Nk, max_nr);
  // (a call that takes another schedule -- Nk > 16, a 5000-node sample, ... -- builds what it needs in its workspace and leaves the
  // external shadows alone: the promise is simply not used)
  if (ext && (reinterpret_cast<uintptr_t>(shadows) & 255)) return fail(CAMO_E_ARG, "the shadow buffer must be 256-byte aligned");
  t_shadows = ext ? shadows : nullptr; t_shadows_valid = ext && shadows_valid != 0;
  t_tail_event = static_cast<hipEvent_t>(tail_event);
  const int rc = forward_loss_backward_impl(dims, params, grads, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace,
                                            workspace_bytes, y, e, s, outs, loss_terms, pred, training, seed, precision, stream);
  t_shadows = nullptr; t_shadows_valid = false;
  if (rc == 0) { CK(record_tail_event(static_cast<hipStream_t>(stream)), "tail event"); }   // (schedules without an early point)
  t_tail_event = nullptr;
  return rc;
}

static int forward_loss_backward_impl(const camo_dims_t* dims, const float* const* params, float* const* grads, const float* rg,
                               const int32_t* rg_offsets, const void* batch_desc, const float* kg,
                               int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace, size_t workspace_bytes,
                               const int64_t* y, const float* e, const float* s, float* outs, float* loss_terms, int32_t* pred,
                               int32_t training, uint64_t seed, int32_t precision, void* stream) {
  const int head0 = dims->fusion_type == CAMO_FUSION_LATE ? CAMO_PL_HEADS : CAMO_P_HEADS;
  const bool fuse = heads_loss_ok(B, dims->num_classes);
  if (g_opt_tail17 != 0 && params && !check_dims(dims, B, T, Nk) && fused17_ok(*dims, params, precision, Nk, max_nr) &&
      tail_fused_ok(B, dims->num_classes)) {
    // fused schedule + one-launch tail: node-level forward, [tail forward + loss + tail backward], node-level backward
    const FusedLoss fl{y, e, s, loss_terms, pred, grads + head0};
    if (int rc = forward_impl(dims, params, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes, outs,
                              nullptr, nullptr, training, seed, precision, 0, stream, nullptr, &fl)) return rc;
    Ws w = carve(*dims, B, T, Nk, workspace);
    bind_shadows(w);
    const Desc bd = desc_carve(B, T, const_cast<void*>(batch_desc));
    CK(record_tail_event(static_cast<hipStream_t>(stream)), "tail event");
    return backward_nodes17(*dims, params, grads, rg_offsets, bd, B, T, Nk, w, make_drop(training, dims->dropout, seed),
                            static_cast<hipStream_t>(stream));
  }
  if (fuse) {
    const FusedLoss fl{y, e, s, loss_terms, pred, grads + head0};
    if (int rc = forward_impl(dims, params, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes, outs,
                              nullptr, nullptr, training, seed, precision, 0, stream, &fl)) return rc;
    return backward_impl(dims, params, grads, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes,
                         outs, nullptr, 1, training, seed, precision, 0, stream, true);
  }
  // large batches / many classes: the three steps as separate launches, d(loss)/d(pre-activation) staged in the workspace
  if (int rc = forward_impl(dims, params, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes, outs,
                            nullptr, nullptr, training, seed, precision, 0, stream, nullptr)) return rc;
  const Ws w = carve(*dims, B, T, Nk, workspace);
  CK(launch_loss(outs, reinterpret_cast<const long long*>(y), e, s, B, dims->num_classes, loss_terms, nullptr, w.dlog, pred,
                 static_cast<hipStream_t>(stream)), "loss");
  return backward_impl(dims, params, grads, rg, rg_offsets, batch_desc, kg, B, T, Nk, max_nr, workspace, workspace_bytes,
                       outs, w.dlog, 1, training, seed, precision, 0, stream, false);
}

int camo_loss(const float* outs, const int64_t* y, const float* e, const float* s, int32_t B, int32_t num_classes,
              float* loss_terms, float* d_outs, float* d_pre, int32_t* pred, void* stream) {
  if (!outs || !y || !e || !s || !loss_terms) return fail(CAMO_E_ARG, "null pointer argument");
  if (B < 1 || num_classes < 1 || num_classes > 64) return fail(CAMO_E_ARG, "need B >= 1 and 1 <= num_classes <= 64");
  CK(launch_loss(outs, reinterpret_cast<const long long*>(y), e, s, B, num_classes, loss_terms, d_outs, d_pre, pred,
                 static_cast<hipStream_t>(stream)), "loss");
  return 0;
}

int camo_grad_sumsq(const float* g, size_t n, float* sumsq, void* stream) {
  if (!g || !sumsq || n == 0) return fail(CAMO_E_ARG, "null pointer or empty buffer");
  CK(launch_sumsq(g, n, sumsq, static_cast<hipStream_t>(stream)), "grad sumsq");
  return 0;
}

size_t camo_shadow_bytes(const camo_dims_t* dims) {
  OptScope opt_scope(dims);
  if (!dims || !fused17_dims(*dims)) return 0;
  return shadow_carve(nullptr).bytes;
}

int camo_clip_adamw_shadows(const camo_dims_t* dims, const float* const* params, float* p, float* g, float* m, float* v, size_t n,
                            float* sumsq, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                            int32_t step, int32_t zero_grads, void* shadows, void* stream) {
  OptScope opt_scope(dims);
  if (!dims || !params || !p || !g || !m || !v || !sumsq || !shadows || n == 0) return fail(CAMO_E_ARG, "null pointer or empty buffer");
  if (step < 1) return fail(CAMO_E_ARG, "step is 1-based");
  if (!fused17_dims(*dims)) return fail(CAMO_E_UNSUPPORTED, "weight shadows exist for the fused schedule's configuration only");
  if (reinterpret_cast<uintptr_t>(shadows) & 255) return fail(CAMO_E_ARG, "the shadow buffer must be 256-byte aligned");
  const int H = 256, D = 128;
  const size_t HH = (size_t)H * H;
  const ShadowSet x = shadow_carve(shadows);
  AdamShadowArgs a; std::memset(&a, 0, sizeof(a));
  struct Cov { size_t off, len; } cov[ADAM_SHADOW_MAXB];
  int ncov = 0;
  bool ok = true;
  auto blk = [&](const float* src, int rows, int cols, us16* plain, int pN, int pn0, us16* trans, int tK, int tk0) {
    if (!src || src < p || src + (size_t)rows * cols > p + n) { ok = false; return; }
    AdamShadowBlock& B = a.blk[a.nblk++];
    B.off = (size_t)(src - p); B.rows = rows; B.cols = cols; B.plain = plain; B.pN = pN; B.pn0 = pn0; B.trans = trans; B.tK = tK; B.tk0 = tk0;
    cov[ncov++] = Cov{B.off, (size_t)rows * cols};
  };
  const float* const* P = params;
  blk(P[CAMO_P_RG_PROJ_W], H, D, x.Wrg, H, 0, nullptr, 0, 0);
  blk(P[CAMO_P_KG_PROJ_W], H, D, x.Wkg, H, 0, nullptr, 0, 0);
  blk(P[CAMO_P_A1_IN_W], H, H, x.Wqkv_rg, 3 * H, 0, x.WcRgT, 3 * H, 0);                    // Wq1
  blk(P[CAMO_P_A1_IN_W] ? P[CAMO_P_A1_IN_W] + HH : nullptr, 2 * H, H, x.Wqkv_kg, 3 * H, H, x.WcKgT, 3 * H, H);   // Wk1 | Wv1
  blk(P[CAMO_P_A2_IN_W], H, H, x.Wqkv_kg, 3 * H, 0, x.WcKgT, 3 * H, 0);                    // Wq2
  blk(P[CAMO_P_A2_IN_W] ? P[CAMO_P_A2_IN_W] + HH : nullptr, 2 * H, H, x.Wqkv_rg, 3 * H, H, x.WcRgT, 3 * H, H);   // Wk2 | Wv2
  blk(P[CAMO_P_A1_OUT_W], H, H, x.Wo1, H, 0, x.Wo1T, H, 0);
  blk(P[CAMO_P_A2_OUT_W], H, H, x.Wo2, H, 0, x.Wo2T, H, 0);
  blk(P[CAMO_P_F1_W0], 2 * H, H, x.W1, 2 * H, 0, x.W1T, 2 * H, 0);
  blk(P[CAMO_P_F2_W0], 2 * H, H, x.W2, 2 * H, 0, x.W2T, 2 * H, 0);
  if (!ok) return fail(CAMO_E_ARG, "the shadowed parameters must lie inside the flat buffer [p, p + n)");
  // the rest of the flat buffer: the gaps between the shadowed blocks, in address order
  for (int i = 1; i < ncov; ++i)
    for (int j = i; j > 0 && cov[j].off < cov[j - 1].off; --j) { const Cov t = cov[j]; cov[j] = cov[j - 1]; cov[j - 1] = t; }
  size_t at = 0;
  for (int i = 0; i <= ncov; ++i) {
    const size_t end = i < ncov ? cov[i].off : n;
    if (end < at) return fail(CAMO_E_ARG, "overlapping parameter blocks");
    if (end > at) {
      if (a.nrange >= ADAM_SHADOW_MAXR) return fail(CAMO_E_ARG, "too many gaps between the shadowed parameters");
      if ((at & 3) || ((end - at) & 3)) return fail(CAMO_E_ARG, "parameters must be 16-byte aligned slices of the flat buffer");
      a.range_begin[a.nrange] = at; a.range_len[a.nrange++] = end - at;
    }
    if (i < ncov) at = cov[i].off + cov[i].len;
  }
  CK(launch_clip_adamw_shadows(p, g, m, v, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, step, zero_grads, a,
                               static_cast<hipStream_t>(stream)), "clip+adamw+shadows");
  return 0;
}

int camo_clip_adamw(float* p, float* g, float* m, float* v, size_t n, float* sumsq, float max_norm, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grads, void* stream) {
  if (!p || !g || !m || !v || !sumsq || n == 0) return fail(CAMO_E_ARG, "null pointer or empty buffer");
  if (step < 1) return fail(CAMO_E_ARG, "step is 1-based");
  CK(launch_clip_adamw(p, g, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, step, zero_grads,
                       static_cast<hipStream_t>(stream)), "clip+adamw");
  return 0;
}

int camo_debug_gemm(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias,
                    const float* res, int32_t ldr, float* bias_grad, int32_t M, int32_t N, int32_t K, int32_t flags,
                    int32_t precision, void* stream) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1) return fail(CAMO_E_ARG, "bad gemm arguments");
  GB g(make_drop(0, 0.f, 0), precision, static_cast<hipStream_t>(stream));
  GemmProb& p = g.add(A, lda, B, ldb, C, ldc, M, N, K, flags);
  p.bias = bias; p.res = res; p.ldr = ldr; p.bias_grad = bias_grad;
  CK(g.run(), "debug gemm");
  return 0;
}

int camo_debug_gemm16(const void* A16, int32_t lda, const void* B16, int32_t ldb, float* C, int32_t ldc, void* C16,
                      int32_t ldc16, const float* bias, const float* res, int32_t ldr, float* bias_grad, int32_t M,
                      int32_t N, int32_t K, int32_t flags, void* stream) {
  if (!A16 || !B16 || (!C && !C16) || M < 1 || N < 1 || K < 1) return fail(CAMO_E_ARG, "bad gemm16 arguments");
  Gemm16Batch gb;
  std::memset(&gb, 0, sizeof(gb));
  gb.drop = make_drop(0, 0.f, 0);
  Gemm16Prob& p = gb.p[0];
  gb.n = 1;
  p.A = static_cast<const unsigned short*>(A16); p.lda = lda; p.B = static_cast<const unsigned short*>(B16); p.ldb = ldb;
  p.C = C; p.ldc = ldc; p.C16 = static_cast<unsigned short*>(C16); p.ldc16 = ldc16;
  p.bias = bias; p.res = res; p.ldr = ldr; p.bias_grad = bias_grad; p.M = M; p.N = N; p.K = K; p.flags = flags; p.aux_scale = 1.f;
  CK(launch_gemm16_batch(gb, static_cast<hipStream_t>(stream)), "debug gemm16");
  return 0;
}

int camo_options_init(camo_options_t* o) {
  if (!o) return fail(CAMO_E_ARG, "options is null");
  *o = k_default_options;
  return 0;
}

int camo_options_set(camo_options_t* o, const char* name, int32_t value) {
  if (!o || !name) return fail(CAMO_E_ARG, "options or option name is null");
  struct Field { const char* name; int32_t camo_options_t::*m; };
  static const Field fields[] = {
      {"sched16", &camo_options_t::sched16}, {"fused", &camo_options_t::fused}, {"tail17", &camo_options_t::tail17}, {"fused_rt", &camo_options_t::fused_rt},
      {"wide2", &camo_options_t::wide2}, {"fused_one", &camo_options_t::fused_one}, {"wide_front_rt", &camo_options_t::wide_front_rt}, {"tailw", &camo_options_t::tailw},
      {"tailw_bwd", &camo_options_t::tailw_bwd}, {"param_space", &camo_options_t::param_space}, {"tn_big", &camo_options_t::tn_big},
      {"fused_variant", &camo_options_t::fused_variant}, {"back_lead", &camo_options_t::back_lead}, {"tn_balance", &camo_options_t::tn_balance},
      {"tn_kcap", &camo_options_t::tn_kcap}, {"tn_exp", &camo_options_t::tn_exp}, {"exp", &camo_options_t::exp}, {"fused_save", &camo_options_t::fused_save},
      {"tail_skip_arrival", &camo_options_t::tail_skip_arrival}, {"wide2_bwd", &camo_options_t::wide2_bwd}};
  for (const Field& f : fields)
    if (std::strcmp(name, f.name) == 0) { o->*(f.m) = value; return 0; }
  return fail(CAMO_E_ARG, std::string("unknown option ") + name);
}

int camo_debug_set_stamps(void* buf, int32_t blocks_per_kernel) {
  g_dbg_stamps = static_cast<unsigned long long*>(buf); g_dbg_stamp_blocks = blocks_per_kernel;
  return 0;
}

int camo_prof_begin(int32_t max_launches) {
  CK(gemm_prof_begin(max_launches), "prof begin");
  return 0;
}

int camo_tail_timeouts(uint32_t* count) {
  if (!count) return fail(CAMO_E_ARG, "null pointer argument");
  unsigned int n = 0;
  CK(tail_timeouts(&n), "tail timeouts");
  *count = n;
  return 0;
}

int camo_tail_poison_to_grads(float* flat_grads, void* stream) {
  if (!flat_grads) return fail(CAMO_E_ARG, "camo_tail_poison_to_grads: null gradient buffer");
  const int e = launch_tail_poison_to_grads(flat_grads, static_cast<hipStream_t>(stream));
  return e ? fail_hip(e, "camo_tail_poison_to_grads") : 0;
}

int camo_prof_kind(int32_t kind, double* ms, int32_t* launches, double* flops) {
  int n = 0;
  CK(gemm_prof_kind(kind, ms, &n, flops), "prof kind");
  if (launches) *launches = n;
  return 0;
}

int camo_prof_end(double* gemm_ms, int32_t* gemm_launches, double* gemm_flops) {
  int n = 0;
  CK(gemm_prof_end(gemm_ms, &n, gemm_flops), "prof end");
  if (gemm_launches) *gemm_launches = n;
  return 0;
}

int64_t camo_debug_ws_offset(const camo_dims_t* dims, int32_t B, int32_t T, int32_t Nk, const char* name) {
  OptScope opt_scope(dims);
  if (check_dims(dims, B, T, Nk) || !name) return -1;
  char* base = reinterpret_cast<char*>(4096);
  const Ws w = carve(*dims, B, T, Nk, base);
  const struct { const char* n; const void* p; } tab[] = {
      {"R16", w.f.R16}, {"G16", w.f.G16}, {"Q16", w.f.Q16}, {"Q2_16", w.f.Q2_16}, {"KV16", w.f.KV16}, {"KV2_16", w.f.KV2_16},
      {"O16", w.f.O16}, {"O2_16", w.f.O2_16}, {"Y16", w.f.Y16}, {"Y2_16", w.f.Y2_16}, {"XH16", w.f.XH16}, {"XH2_16", w.f.XH2_16},
      {"rstd1", w.f.rstd1}, {"rstd2", w.f.rstd2}, {"mask1", w.f.mask1}, {"mask2", w.f.mask2}, {"lse2", w.f.lse2}, {"X16", w.f.X16},
      {"Wqkv_rg", w.f.Wqkv_rg}, {"W1s", w.f.W1}, {"W1T", w.f.W1T}, {"WcRgT", w.f.WcRgT}, {"dH16", w.f.dH16}, {"dH2_16", w.f.dH2_16},
      {"dU16", w.f.dU16}, {"dU2_16", w.f.dU2_16}, {"dQKV16", w.f.dQKV16}, {"dQKVkg16", w.f.dQKVkg16}, {"dR16", w.f.dR16}, {"dG16", w.f.dG16},
      {"dO2_16", w.f.dO2_16}, {"delta2", w.f.delta2}, {"dKV", w.dKV}, {"dQ2acc", w.dQ2acc}, {"Ymean", w.Ymean}, {"H1mean", w.H1mean}, {"Y2mean", w.Y2mean}, {"H2mean", w.H2mean},
      {"R", w.R}, {"G", w.G}, {"Q", w.Q}, {"KV2", w.KV2}, {"KV", w.KV}, {"Q2", w.Q2}, {"P", w.P}, {"P2", w.P2},
      {"O", w.O}, {"O2", w.O2}, {"U", w.U}, {"U2", w.U2}, {"Y", w.Y}, {"Y2", w.Y2}, {"H1", w.H1}, {"H2", w.H2},
      {"comb", w.comb}, {"fused", w.fused}, {"F1", w.F1}, {"hid", w.hid}, {"dhid", w.dhid}, {"dfused", w.dfused}, {"dF1", w.dF1},
      {"dcomb", w.dcomb}, {"dHm1", w.dHm1}, {"dHm2", w.dHm2}};
  for (const auto& e : tab)
    if (std::strcmp(e.n, name) == 0 && e.p) return static_cast<const char*>(e.p) - base;
  return -1;
}


// ---- Region-Graph GNN embedding path (include/camo_rg_gnn.h) ----------------------------------------------
namespace {
struct RgWs { float *Hh, *a_src, *a_dst, *dinv, *xw, *ha, *hb; size_t bytes; };
RgWs rg_carve(const camo_rg_dims_t& d, int N, void* base) {
  RgWs w{};
  Carver c(base);
  const size_t n = N, C = d.hidden, K = d.heads;
  w.Hh = c.take<float>(n * K * C); w.a_src = c.take<float>(n * K); w.a_dst = c.take<float>(n * K); w.dinv = c.take<float>(n);
  w.xw = c.take<float>(n * C); w.ha = c.take<float>(n * C); w.hb = c.take<float>(n * C);
  c.off = (c.off + 255) & ~size_t(255);
  w.bytes = c.off;
  return w;
}
int rg_check(const camo_rg_dims_t* d, int N) {
  if (!d) return fail(CAMO_E_ARG, "dims is null");
  if (N < 1 || d->in_channels < 1 || d->hidden < 1 || d->hidden > 512 || d->heads < 1 || d->heads > 8)
    return fail(CAMO_E_UNSUPPORTED, "need N >= 1, hidden <= 512, 1 <= heads <= 8");
  return 0;
}
}  // namespace

int camo_rg_build_csr(const int64_t* edge_index, const float* edge_weight, int32_t N, int32_t E, int32_t* scratch, int32_t* rowptr,
                      int32_t* col, float* w, void* stream) {
  if (N < 1 || E < 0 || (E > 0 && !edge_index) || !scratch || !rowptr || !col || !w) return fail(CAMO_E_ARG, "bad build_csr arguments");
  // scratch: 3 N words = counts | cursor | self-loop weights
  CK(launch_build_csr(reinterpret_cast<const long long*>(edge_index), reinterpret_cast<const long long*>(edge_index) + E, edge_weight, N, E,
                      scratch, reinterpret_cast<float*>(scratch + 2 * (size_t)N), scratch + N, rowptr, col, w, static_cast<hipStream_t>(stream)),
     "build csr");
  return 0;
}

size_t camo_rg_workspace_bytes(const camo_rg_dims_t* dims, int32_t N) {
  if (rg_check(dims, N)) return 0;
  return rg_carve(*dims, N, nullptr).bytes;
}

int camo_rg_node_embeddings(const camo_rg_dims_t* dims, const float* const* params, const float* x, const int32_t* rowptr,
                            const int32_t* col, const float* w, int32_t N, int32_t E, void* workspace, size_t workspace_bytes,
                            float* out, void* stream) {
  if (int e = rg_check(dims, N)) return e;
  if (!params || !x || !rowptr || !col || !w || !workspace || !out || E < N) return fail(CAMO_E_ARG, "null pointer argument or E < N (one self-loop per node)");
  const camo_rg_dims_t& d = *dims;
  const RgWs ws = rg_carve(d, N, workspace);
  if (workspace_bytes < ws.bytes) return fail(CAMO_E_WORKSPACE, "workspace smaller than camo_rg_workspace_bytes()");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int C = d.hidden, K = d.heads, In = d.in_channels;
  const float* const* P = params;
  auto bn = [&](int slot) { return BnEval{P[slot], P[slot + 1], P[slot + 2], P[slot + 3]}; };
  GB g(make_drop(0, 0.f, 0), CAMO_PREC_F32, st);
  // conv1: GATConv (extract_rg_embeddings.py:104) + bn1 + relu
  g.nt(x, In, P[CAMO_RG_C1_W], In, nullptr, ws.Hh, K * C, N, K * C, In);
  CK(g.run(), "gat projection");
  CK(launch_gat_alpha(ws.Hh, P[CAMO_RG_C1_ATT_SRC], P[CAMO_RG_C1_ATT_DST], ws.a_src, ws.a_dst, N, K, C, st), "gat attention logits");
  CK(launch_gat_aggregate(ws.Hh, ws.a_src, ws.a_dst, rowptr, col, P[CAMO_RG_C1_BIAS], bn(CAMO_RG_BN1), ws.ha, N, K, C, st), "gat aggregate");
  // conv2..4: GCNConv with edge weights (:108-118) + bn + relu
  CK(launch_gcn_dinv(rowptr, w, ws.dinv, N, st), "gcn degrees");
  float* cur = ws.ha; float* nxt = ws.hb;
  for (int k = 0; k < 3; ++k) {
    const int base = CAMO_RG_C2_BIAS + 6 * k;
    g.nt(cur, C, P[base + 1], C, nullptr, ws.xw, C, N, C, C);
    CK(g.run(), "gcn projection");
    CK(launch_gcn_aggregate(ws.xw, rowptr, col, w, ws.dinv, P[base], bn(base + 2), nxt, N, C, st), "gcn aggregate");
    float* t = cur; cur = nxt; nxt = t;
  }
  // fc_shared + relu (:121)
  g.nt(cur, C, P[CAMO_RG_FC_W], C, P[CAMO_RG_FC_B], out, C, N, C, C, GF_RELU);
  CK(g.run(), "fc_shared");
  return 0;
}

size_t camo_rg_graph_workspace_bytes(int32_t n_labels) {
  if (n_labels < 1 || n_labels > CAMO_RG_MAX_LABELS) return 0;
  return rg_graph_carve(n_labels, nullptr).bytes;
}

int camo_rg_region_graph(const float* image, const int32_t* segments, const uint8_t* canny, int32_t H, int32_t W, int32_t n_labels,
                         void* workspace, size_t workspace_bytes, float* x, int32_t* region_map, int64_t* edge_index,
                         float* edge_attr, int32_t edge_capacity, int32_t* counts, void* stream) {
  if (!image || !segments || !canny || !workspace || !x || !region_map || !edge_index || !edge_attr || !counts)
    return fail(CAMO_E_ARG, "null pointer argument");
  if (H < 1 || W < 1 || (long long)H * W > (1ll << 26)) return fail(CAMO_E_ARG, "image size out of range");
  if (n_labels < 1 || n_labels > CAMO_RG_MAX_LABELS) return fail(CAMO_E_ARG, "n_labels must be in [1, 4096]");
  if (edge_capacity < 2) return fail(CAMO_E_ARG, "edge_capacity must be >= 2");
  const RgGraphWs ws = rg_graph_carve(n_labels, workspace);
  if (workspace_bytes < ws.bytes) return fail(CAMO_E_WORKSPACE, "workspace smaller than camo_rg_graph_workspace_bytes()");
  CK(launch_region_graph(image, segments, canny, H, W, n_labels, ws, x, region_map, reinterpret_cast<long long*>(edge_index), edge_attr,
                         edge_capacity, counts, static_cast<hipStream_t>(stream)), "region graph");
  return 0;
}
}  // extern "C"
