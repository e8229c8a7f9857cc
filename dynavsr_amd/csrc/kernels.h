// Internal C++ launch API shared by the C-ABI wrappers and the EDVR engine.
#pragma once
#include "../../include/dynavsr_hip.h"
#include "common.h"

namespace dvsr {

// Extra modes of the conv kernel used by backward-data (dgrad) launches.
struct ConvExtra {
  int wt = 0;                  // 1: weights read as the transposed, tap-mirrored view (dgrad)
  int w_ctot = 0, w_coff = 0;  // wt=1: original conv's total input channels / channel offset of this input
  int in_ps = 0;               // input is stored pixel-shuffled (gradient of a PixelShuffle(2) output)
  int in_dil = 0, Hs = 0, Ws = 0;  // input is the zero-dilated view of a [N][c0][Hs][Ws] tensor
  int accum = 0;               // y += result
  const float* gmask = nullptr;  // dgrad: multiply by act'(gmask) (fused activation backward of the producer)
  int gmask_act = 0;
  // per-sample weight sets: batch item n takes the packed weights at + (n / wdiv) * w_gs floats, the bias at + (n / wdiv) * b_gs
  int wdiv = 1; long long w_gs = 0; int b_gs = 0;
};
int conv2d_run(const dvsr_conv2d_desc& d, const ConvExtra& ex, hipStream_t st);

int mdcn_forward_run(const float* x, const float* off, long long off_bs, const float* msk,
                     long long msk_bs, int mask_logit, const float* w, const float* b, float* out,
                     int N, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad,
                     int dil, int groups, int dg, int act, hipStream_t st);

// conv2d_v2.hip: pipelined kernel over pre-packed weights
struct PackEntry {
  const float* w; float* P;
  int Cout, Ctot, KK, CC, wt, w_ctot, w_coff, ncb, nchunks, pch;
  int bf = 0;  // 1: bf16 image for the bf16 MFMA kernel (pch still counts fp32-sized slots)
  int perm = 0;  // 1: channel order of the DMA-halo kernel (ConvGeo::dma); 3: Winograd-transformed image (conv2d_wino.hip);
                 // 4: the same as three bf16 pieces (conv2d_wino3.hip); 5: the F(4x4, 3x3) image of conv2d_wino5.hip
};
struct PackTable {
  int n;
  PackEntry e[48];
};
int conv2_pch(int ks, int stride);  // floats per packed (cout block, chunk)
int conv2_cc(int ks, int stride);   // input channels per chunk
int pack_weights_run(const PackTable& t, hipStream_t st);
// channels per chunk, tile rows (x32 px), 32-cout halves per workgroup; dma: 1 DMA-halo kernel, 2 row-split 7x7 / 9x9,
// 3 Winograd F(2x2, 3x3) kernel (th = 4: 4x64-pixel tiles, th = 8: 8x32), 4 the same on the bf16 pipe (exact 3-way split),
// 5 Winograd F(4x4, 3x3) on the bf16 pipe (conv2d_wino5.hip; th = 8: 8x64-pixel workgroup tiles, th = 16: 16x32)
struct ConvGeo { int cc, th, mt; int bf = 0; int dma = 0; };
// allow: bit 0 = the K-split small-grid kernel may be chosen, bit 1 = the DMA-halo kernel (plain pad-1 inputs, see conv2d_v2.hip),
// bit 2 = the Winograd kernel (bit 1's conditions + an epilogue it implements: plain or PixelShuffle(2) stores)
ConvGeo conv2_choose(int ks, int stride, int N, int Ho, int Wo, int Cout, int Ctot, int allow_ksplit = 1);
int conv2_pch_cc(int ks, int cc, int bf = 0, int dma = 0);   // fp32-sized slots per packed (64-cout block, chunk of cc channels)
int pack_weights_wino_run(const PackTable& t, hipStream_t st);   // conv2d_wino.hip: entries with perm == 3
int pack_weights_wino3_run(const PackTable& t, hipStream_t st);  // conv2d_wino3.hip: entries with perm == 4
int pack_weights_wino5_run(const PackTable& t, hipStream_t st);  // conv2d_wino5.hip: entries with perm == 5 (F(4x4, 3x3))
int conv2d_packed_run(const dvsr_conv2d_desc& d, const float* wp, const ConvExtra& ex, const ConvGeo& geo,
                      hipStream_t st);

// conv1x1_dual.hip: two 1x1 convolutions (64 outputs each) over one input in one pass; x [N][Cin][HW], w [64][Cin]
bool conv1x1_dual_ok(const float* x, const float* w0, const float* w1, const float* y0, const float* y1, int Cin, long long HW);
int conv1x1_dual_run(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* y0, float* y1,
                     int N, int Cin, int HW, int act, hipStream_t st, int wdiv = 1, long long w_gs = 0, int b_gs = 0);
int conv3x3_small_cout_run(const float* x, const float* w, const float* bias, const float* res, float* y, int N,
                           int C, int H, int W, int Cout, int act, hipStream_t st, int wdiv = 1, long long w_gs = 0,
                           int b_gs = 0);

// Deferred slot reduction of a weight gradient (conv2d_wgrad_run(..., defer = &entry) + wgrad_reduce_batch)
struct WgradReduceEntry {
  float* partial; float* dbp; float* dW; float* db;
  int nslot, KK, OP, CP, Cout, Cin, Ctot, c_off;
  // per-group gradients: group g reads the slots [g * nslot, (g + 1) * nslot) and writes dW + g * dW_gs, db + g * db_gs
  int ngroups = 1;
  long long dW_gs = 0, db_gs = 0;
};
constexpr int WGRAD_REDUCE_BATCH = 44;  // entries per launch (kernel-argument limit: 44 x 88 B < 4 KB)
struct WgradReduceTable {
  int n;
  WgradReduceEntry e[WGRAD_REDUCE_BATCH];
};
int wgrad_reduce_batch(const WgradReduceEntry* entries, int n, hipStream_t st);
size_t conv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad = -1, int groups = 1);
// groups > 1: one gradient per group of N / groups consecutive batch items, written to dW + g * dW_gs (db + g * db_gs)
int conv2d_wgrad_run(const float* x, long long x_bs, int x_bdiv, const float* gy, int gy_ps, float* dW,
                     float* db, int N, int Cin, int H, int W, int Cout, int Ctot, int c_off, int ks,
                     int stride, void* ws, size_t ws_bytes, hipStream_t st, int scratch_is_zero = 0,
                     int pad = -1, WgradReduceEntry* defer = nullptr, int groups = 1, long long dW_gs = 0,
                     long long db_gs = 0);  // pad < 0: ks / 2
size_t mdcn_backward_workspace_bytes(int N, int C, int H, int W, int Cout, int stride, int pad, int dil, int groups = 1);
int mdcn_backward_run(const float* x, const float* off, long long off_bs, const float* msk, long long msk_bs,
                      int mask_logit, const float* w, const float* gout, float* gx, float* goff,
                      long long goff_bs, float* gmsk, long long gmsk_bs, float* gw, float* gb, int N, int C,
                      int H, int W, int Cout, int stride, int pad, int dil, int dg, void* ws,
                      size_t ws_bytes, hipStream_t st, int groups = 1, long long gw_gs = 0, long long gb_gs = 0,
                      long long w_gs = 0);  // w_gs != 0: group g of the batch convolves with w + g * w_gs (per-sample weights)

// pack_perm: the layout `wp` is in -- 0 the fp32 conv pack (pack_weights_kernel, 8-channel chunks), 6 the three bf16 pieces of
// mdcn_split.hip; whoever makes the pack asks mdcn_pack_perm(W) and hands the answer back here.
int mdcn_forward_packed_run(const float* x, const float* off, long long off_bs, const float* msk,
                            long long msk_bs, int mask_logit, const float* wp, const float* b, float* out,
                            int N, int C, int H, int W, int Cout, int dg, int act, hipStream_t st, int wdiv = 1,
                            long long w_gs = 0, int b_gs = 0, int pack_perm = 0);
int mdcn_fwd_variant();      // DVSR_DCN_FWD, read once: 3 split (default), 0 dma, 2 reg
int mdcn_pack_floats();      // fp32-sized slots per (64-cout block, 8-channel chunk) of a DCN weight pack (either layout fits)
int mdcn_pack_perm(int W);   // PackEntry::perm of a DCN weight pack for images of width W
int pack_weights_dcn3_run(const PackTable& t, hipStream_t st);   // mdcn_split.hip: entries with perm == 6

struct DcnK2 {
  const float* x; const float* off; const float* msk; const float* wp; const float* bias; float* out;
  long long off_bstride, msk_bstride;
  int mask_logit;
  int N, C, H, W, Cout, dg, act;
  int tiles_x, tiles_y, ntiles, ncb, nchunks;
  int wdiv = 1; long long w_gs = 0; int b_gs = 0;   // per-sample weight sets (common.h: wset_ptr)
#ifdef DVSR_CONV_TRACE
  long long* trace;  // debug build only (tools/dcn_trace.py): 64 cycle stamps per workgroup
#endif
};
int mdcn_fwd_split_launch(const DcnK2& k, int grid, int mask_logit, hipStream_t st);   // mdcn_split.hip

// misc.hip
int upsample_bilinear_fwd(const float* x, float* y, size_t planes, int H, int W, int S, float mul,
                          hipStream_t st);
int upsample_bilinear_bwd(const float* gy, float* gx, size_t planes, int H, int W, int S, float mul,
                          int accumulate, hipStream_t st);
int pool3s2_fwd(const float* x, float* ymax, float* yavg, size_t planes, int H, int W, hipStream_t st);
int pool3s2_bwd(const float* x, const float* gmax, const float* gavg, float* gx, size_t planes,
                int H, int W, int accumulate, hipStream_t st);
int reduce_frames(float* dst, long long dst_bs, const float* src, int B, int cnt, size_t per,
                  int accumulate, hipStream_t st);
int tsa_gate_fwd(const float* emb, const float* emb_ref, const float* aligned, float* cor,
                 float* gated, int B, int N, int C, size_t HW, hipStream_t st);
int tsa_gate_bwd(const float* emb, const float* emb_ref, const float* aligned, const float* cor,
                 const float* g_gated, float* g_emb, float* g_emb_ref, float* g_aligned, int B,
                 int N, int C, size_t HW, hipStream_t st, float* gdot_scratch = nullptr);
int tsa_blend_fwd(const float* fea, const float* att, const float* add, float* out, size_t n,
                  hipStream_t st);
int tsa_blend_bwd(const float* fea, const float* att, const float* g, float* g_fea, float* g_att_io,
                  size_t n, int accumulate, hipStream_t st);
int add_inplace(float* dst, const float* src, size_t n, hipStream_t st);
int add_out(float* y, const float* a, const float* b, size_t n, hipStream_t st);
int act_bwd_inplace(float* g, const float* y, size_t n, int act, hipStream_t st);

// pad.hip: explicit padding / layout changes of the MFDN estimator and their adjoints
enum : int { PAD_REFLECT = 0, PAD_REFLECT_S2D = 1, PAD_REPL_T3 = 2 };
size_t pad_out_numel(int mode, size_t N, int C, int H, int W);
int pad_fwd(const float* x, float* y, int mode, int N, int C, int H, int W, int T, hipStream_t st);
int pad_bwd(const float* gy, float* gx, int mode, int N, int C, int H, int W, int T, int accumulate,
            hipStream_t st, const float* gmask = nullptr, int gmask_act = 0, int gmask_padded = 0);
int meansub_slices();  // scratch floats per plane
int meansub_fwd(const float* x, float* xm, float* mean, float* part, int B, int C, int T, int H, int W,
                hipStream_t st);
int addmean_fwd(const float* y, const float* mean, float* out, int B, int C, int T, size_t HW, hipStream_t st);
int addmean_bwd(const float* gout, float* gy, int B, int C, int T, size_t HW, hipStream_t st);
int w4_to_s2d(const float* w, float* w2, int Cout, int C, int inverse, hipStream_t st);

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// Sampling geometry of one (pixel, tap) of the deformable conv: 4 corner offsets and weights;
// an invalid corner contributes 0 (deform_conv_cuda_kernel.cu:479-490).
struct DcnTap {
  int o1, o2, o3, o4;    // element offsets inside a plane (0 when the corner is invalid)
  float w1, w2, w3, w4;  // hh*hw, hh*lw, lh*hw, lh*lw
  float lh, lw;
  bool v1, v2, v3, v4;
};

// false <=> the sample is outside the (-1,H)x(-1,W) gate (kernel.cu:617) and contributes nothing.
__device__ __forceinline__ bool make_tap(float h_im, float w_im, int H, int W, DcnTap& t) {
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return false;
  const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
  const int h_hi = h_lo + 1, w_hi = w_lo + 1;
  t.lh = h_im - (float)h_lo;
  t.lw = w_im - (float)w_lo;
  const float hh = 1.f - t.lh, hw = 1.f - t.lw;
  t.v1 = h_lo >= 0 && w_lo >= 0;
  t.v2 = h_lo >= 0 && w_hi <= W - 1;
  t.v3 = h_hi <= H - 1 && w_lo >= 0;
  t.v4 = h_hi <= H - 1 && w_hi <= W - 1;
  t.o1 = t.v1 ? h_lo * W + w_lo : 0;
  t.o2 = t.v2 ? h_lo * W + w_hi : 0;
  t.o3 = t.v3 ? h_hi * W + w_lo : 0;
  t.o4 = t.v4 ? h_hi * W + w_hi : 0;
  t.w1 = hh * hw; t.w2 = hh * t.lw; t.w3 = t.lh * hw; t.w4 = t.lh * t.lw;
  return true;
}

}  // namespace dvsr
