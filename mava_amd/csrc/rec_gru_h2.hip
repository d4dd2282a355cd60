// GRU sequence scans with the recurrent products on split-f16 ("f16x2") operands
// (rec_gru.hip holds the exact-f32 scans, the reference citations and the ABI entry points that dispatch here).
//
// Same mapping as the f32 scans - one 256-thread block owns 32 sequences for all T steps, wave w owns hidden features
// [32w, 32w+32) of every gate and keeps its slice of W_h in registers - with three changes:
//  * the slice is held as split fragments (hi + lo f16, error diffusion along the summation index; the same 192
//    registers) and a step's recurrent product is 72 v_mfma_f32_32x32x16_f16 per wave instead of 192 f32 MFMAs
//    (2 304 instead of 12 288 matrix-pipe cycles);
//  * the hidden state (forward) / the gate gradients (backward) cross the waves as an LDS IMAGE [row][feature] of
//    f16 hi and lo planes, split ONCE by the lane that produced the value; a consumer's operand fragment (8 consecutive
//    features of its row) is one ds_read_b128 per plane;
//  * the image is double-buffered, so a step has one block barrier instead of two; the f32 hidden state a lane needs for
//    h' = (1-z) n + z h is the one it produced itself and stays in its registers.
// (Measured and rejected: the products the other way round - lane = feature, the 16 accumulator registers = 16 rows, so
// that every T32 access is 16 bytes and a step needs 36 / 48 memory instructions instead of 144 / 192.  Each such
// instruction touches 32 separate 128-byte lines for 32 bytes each; the texture path handles lines, not bytes, and the
// forward scan went from 0.52 to 0.69 ms (critic-sized launch).  The four-byte accesses below cover two FULL lines per
// instruction.)
// Backward: the gate gradients must sit in f16's range - the caller runs the backward chain in units of a power of two
// near the row count (mava_seq_*_loss_f32 grad_scale).
#include "h2_core.h"
#include "rec_task.h"

namespace {

using h2::Frag;
using h2::half4;
using h2::u8;

constexpr int G3 = 3 * MLP_H;
constexpr int HROW = h2::IMG_ROW;           // 272: forward image row (128 f16 + 16)
constexpr int HPLANE = 32 * HROW;
constexpr int HIMG = 2 * HPLANE;
constexpr int GROW = 2 * G3 + 16;           // 784: backward image row (384 f16 + 16; an odd number of 16-byte slots)
constexpr int GPLANE = 32 * GROW;
constexpr int GIMG = 2 * GPLANE;

#define OFFW(r) ((((r) & 3) + 8 * ((r) >> 2)) * 32)

#ifdef MAVA_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                    \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_prev;                                  \
    st_prev = st_now;                                               \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
__device__ unsigned long long* g_scan_stamps = nullptr;
#else
#define STAMP_DECL
#define STAMP(i)
#endif

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// 16 values of a lane (register q <-> feature col0 + (q & 3) + 8 (q >> 2), row r) -> split -> image row r
template <int ROWB, int PLANE>
__device__ __forceinline__ void put16(u8* img, int r, int col0, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    half4 ph, pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a, b;
      h2::split1(v[4 * g + e], a, b);
      ph[e] = a;
      pl[e] = b;
    }
    u8* p = img + r * ROWB + 2 * (col0 + 8 * g);
    *reinterpret_cast<half4*>(p) = ph;
    *reinterpret_cast<half4*>(p + PLANE) = pl;
  }
}

// The reset flags of one sequence, 32 time steps per word.  A flag read inside the time loop sits behind an index load and
// (memory operations retire in order) behind every store of the step before it: it stalled each step for thousands of
// cycles.  Loaded as a chunk it costs one round trip per 32 steps.  base = address of the flag of step 0, stride = E * A.
__device__ __forceinline__ uint32_t done_chunk(const uint8_t* base, long stride, int t0, int T) {
  uint8_t f[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int t = t0 + i;
    f[i] = base[(long)(t < T ? t : (T - 1)) * stride];
  }
  uint32_t bits = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) bits |= (f[i] != 0 && t0 + i < T) ? (1u << i) : 0u;
  return bits;
}

__global__ __launch_bounds__(256, 1) void gru_scan_fwd_h2_kernel(ScanTask tk) {
  __shared__ __attribute__((aligned(16))) u8 IMG[2 * HIMG];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int mt = blockIdx.x;
  const int m = mt * 32 + j;
  const int tiles_per_t = tk.Rm / 32;
  const int fb = 32 * w + 4 * h;

  // resident slice of W_h: fragment (g, b) element e = Wh[16b + 8h + e][g*128 + 32w + j]
  Frag wf[3][8];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    float carry = 0.0f;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tk.wh[(long)(16 * b + 8 * h + e) * G3 + g * MLP_H + 32 * w + j];
      wf[g][b] = h2::split8_carry(v, carry);
    }
  }
  float bn[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bn[r] = tk.bhn[fb + (r & 3) + 8 * (r >> 2)];

  // masked hidden state entering the current step: this lane's 16 (feature, row) elements
  const long row0 = ext_row(tk, 0, m);                 // (the env id behind it is loaded once)
  const uint8_t* const done0 = tk.done + row0;
  const long dstride = (long)tk.E * tk.A;
  uint32_t dbits = done_chunk(done0, dstride, 0, tk.T);  // flags of steps 32c .. 32c + 31 of the current chunk
  float hp[16];
  {
    const bool rs = (dbits & 1u) != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fb + (r & 3) + 8 * (r >> 2);
      float v;
      if (tk.h0_t32) v = tk.h0[((long)mt * MLP_H + f) * 32 + j];
      else v = tk.h0[row0 * MLP_H + f];
      hp[r] = rs ? 0.0f : v;
    }
    put16<HROW, HPLANE>(IMG, j, fb, hp);
  }

  float gr[16], gz[16], gin[16];
  const int lane_off = fb * 32 + j;
  auto load_gi = [&](int t) {
    const float* git = tk.gi + ((long)t * tiles_per_t + mt) * G3 * 32 + lane_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gr[r] = git[OFFW(r)];
      gz[r] = git[MLP_H * 32 + OFFW(r)];
      gin[r] = git[2 * MLP_H * 32 + OFFW(r)];
    }
  };
  load_gi(0);
  STAMP_DECL
  for (int t = 0; t < tk.T; ++t) {
    STAMP(0);
    const long tile = (long)t * tiles_per_t + mt;
    const u8* const img = IMG + (t & 1) * HIMG;
    u8* const img_next = IMG + ((t + 1) & 1) * HIMG;
    f32x16 ar, az, an;
    float gn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ar[r] = gr[r]; az[r] = gz[r]; gn[r] = gin[r]; an[r] = bn[r]; }
    if (((t + 1) & 31) == 0 && t + 1 < tk.T) dbits = done_chunk(done0, dstride, t + 1, tk.T);
    const bool rs_next = (t + 1 < tk.T) ? (((dbits >> ((t + 1) & 31)) & 1u) != 0) : false;
    if (t + 1 < tk.T) load_gi(t + 1);  // in flight during the step
    STAMP(1);
    __syncthreads();  // image of h entering step t complete (and every wave is past its reads of the other buffer)
    STAMP(2);
    {
      const int ro = j * HROW + 16 * h;  // features 8h .. 8h + 7 of row j; batch b adds 32 bytes
      Frag hf[2];
      hf[0] = h2::read_row_frag(img, HPLANE, ro);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b + 1 < 8) hf[(b + 1) & 1] = h2::read_row_frag(img, HPLANE, ro + 32 * (b + 1));
        ar = h2::mfma3(wf[0][b], hf[b & 1], ar);
        az = h2::mfma3(wf[1][b], hf[b & 1], az);
        an = h2::mfma3(wf[2][b], hf[b & 1], an);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(3);
    float hn[16];
    float* const hs_o = tk.hs + tile * (MLP_H * 32) + lane_off;
    if (tk.saved != nullptr && tk.hprev != nullptr) {  // training: one uniform branch, straight-line stores
      float* const hp_o = tk.hprev + tile * (MLP_H * 32) + lane_off;
      float* const sv = tk.saved + tile * (4 * MLP_H) * 32 + lane_off;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rr = sigmoidf_(ar[r]);
        const float zz = sigmoidf_(az[r]);
        const float nn = tanhf_(gn[r] + rr * an[r]);
        hn[r] = (1.0f - zz) * nn + zz * hp[r];
        hs_o[OFFW(r)] = hn[r];
        hp_o[OFFW(r)] = hp[r];
        sv[OFFW(r)] = rr;
        sv[MLP_H * 32 + OFFW(r)] = zz;
        sv[2 * MLP_H * 32 + OFFW(r)] = nn;
        sv[3 * MLP_H * 32 + OFFW(r)] = an[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rr = sigmoidf_(ar[r]);
        const float zz = sigmoidf_(az[r]);
        const float nn = tanhf_(gn[r] + rr * an[r]);
        hn[r] = (1.0f - zz) * nn + zz * hp[r];
        hs_o[OFFW(r)] = hn[r];
      }
    }
    STAMP(4);
#pragma unroll
    for (int r = 0; r < 16; ++r) hp[r] = rs_next ? 0.0f : hn[r];
    put16<HROW, HPLANE>(img_next, j, fb, hp);
    STAMP(5);
  }
#ifdef MAVA_STAMPS
  if (g_scan_stamps != nullptr && blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) g_scan_stamps[i] = st_acc[i];
#endif
}

template <bool NONLY>
__global__ __launch_bounds__(256, 1) void gru_scan_bwd_h2_kernel(ScanTask tk) {
  extern __shared__ __attribute__((aligned(16))) u8 DIMG[];  // 2 x GIMG
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int mt = blockIdx.x;
  const int m = mt * 32 + j;
  const int tiles_per_t = tk.Rm / 32;
  const int fb = 32 * w + 4 * h;

  // resident rows of W_h: fragment b element e = Wh[32w + j][16b + 8h + e]  (sums over the 384 gate columns)
  Frag wb[24];
  {
    float carry = 0.0f;
#pragma unroll
    for (int b = 0; b < 24; ++b) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tk.wh[(long)(32 * w + j) * G3 + 16 * b + 8 * h + e];
      wb[b] = h2::split8_carry(v, carry);
    }
  }
  float dhc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dhc[r] = 0.0f;

  float i_r[16], i_z[16], i_n[16], i_hl[16], i_hp[16], i_dh[16];
  const int lane_off = fb * 32 + j;
  auto load_step = [&](int t) {
    const long tile = (long)t * tiles_per_t + mt;
    const float* sv = tk.saved + tile * (4 * MLP_H) * 32 + lane_off;
    const float* hpv = tk.hprev + tile * (MLP_H * 32) + lane_off;
    const float* dho = tk.dh_out + tile * (MLP_H * 32) + lane_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      i_r[r] = sv[OFFW(r)];
      i_z[r] = sv[MLP_H * 32 + OFFW(r)];
      i_n[r] = sv[2 * MLP_H * 32 + OFFW(r)];
      i_hl[r] = sv[3 * MLP_H * 32 + OFFW(r)];
      i_hp[r] = hpv[OFFW(r)];
      i_dh[r] = dho[OFFW(r)];
    }
  };
  load_step(tk.T - 1);
  const uint8_t* const done0 = tk.done + ext_row(tk, 0, m);
  const long dstride = (long)tk.E * tk.A;
  uint32_t dbits = done_chunk(done0, dstride, (tk.T - 1) & ~31, tk.T);
  for (int t = tk.T - 1; t >= 0; --t) {
    const long tile = (long)t * tiles_per_t + mt;
    u8* const img = DIMG + (t & 1) * GIMG;
    if ((t & 31) == 31 && t != tk.T - 1) dbits = done_chunk(done0, dstride, t & ~31, tk.T);
    const bool rs = ((dbits >> (t & 31)) & 1u) != 0;
    float dhp[16], g_r[16], g_z[16], g_n[16];
    float* const gi_o = tk.dgi + tile * G3 * 32 + lane_off;
    // dgh's r and z thirds are dgi's (the gates add the two pre-activations): NONLY stores the n third alone, 10 instead
    // of 12 row-steps x 512 bytes of traffic per step of this HBM-bound kernel; the consumer reads the thirds where they are
    float* const gh_o = tk.dgh + tile * (NONLY ? MLP_H : G3) * 32 + lane_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float rr = i_r[r], zz = i_z[r], nn = i_n[r], hl = i_hl[r], hp = i_hp[r];
      const float dh = i_dh[r] + dhc[r];
      const float dn = dh * (1.0f - zz);
      const float dz = dh * (hp - nn);
      dhp[r] = dh * zz;
      const float dn_pre = dn * (1.0f - nn * nn);
      const float dr = dn_pre * hl;
      const float dghn = dn_pre * rr;
      const float dz_pre = dz * zz * (1.0f - zz);
      const float dr_pre = dr * rr * (1.0f - rr);
      gi_o[OFFW(r)] = dr_pre;
      gi_o[MLP_H * 32 + OFFW(r)] = dz_pre;
      gi_o[2 * MLP_H * 32 + OFFW(r)] = dn_pre;
      if constexpr (NONLY) {
        gh_o[OFFW(r)] = dghn;
      } else {
        gh_o[OFFW(r)] = dr_pre;
        gh_o[MLP_H * 32 + OFFW(r)] = dz_pre;
        gh_o[2 * MLP_H * 32 + OFFW(r)] = dghn;
      }
      g_r[r] = dr_pre;
      g_z[r] = dz_pre;
      g_n[r] = dghn;
    }
    put16<GROW, GPLANE>(img, j, fb, g_r);
    put16<GROW, GPLANE>(img, j, MLP_H + fb, g_z);
    put16<GROW, GPLANE>(img, j, 2 * MLP_H + fb, g_n);
    if (t > 0) load_step(t - 1);  // in flight during the MFMAs below
    __syncthreads();  // gate-gradient image of step t complete (every wave is past its reads of the other buffer)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
      const int ro = j * GROW + 16 * h;
      Frag gf[2];
      gf[0] = h2::read_row_frag(img, GPLANE, ro);
#pragma unroll
      for (int b = 0; b < 24; ++b) {
        if (b + 1 < 24) gf[(b + 1) & 1] = h2::read_row_frag(img, GPLANE, ro + 32 * (b + 1));
        acc = h2::mfma3(wb[b], gf[b & 1], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // gradient into the hidden state entering step t; a reset cuts the chain (networks.py:253-257)
#pragma unroll
    for (int r = 0; r < 16; ++r) dhc[r] = rs ? 0.0f : (acc[r] + dhp[r]);
  }
}

}  // namespace

#ifdef MAVA_STAMPS
extern "C" int mava_debug_set_scan_stamps(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_stamps), &p, sizeof(p));
}
#endif

int mava_gru_scan_fwd_h2_launch(const ScanTask& tk, hipStream_t s) {
  hipLaunchKernelGGL(gru_scan_fwd_h2_kernel, dim3(tk.Rm / 32), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

int mava_gru_scan_bwd_h2_launch(const ScanTask& tk, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)gru_scan_bwd_h2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GIMG));
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)gru_scan_bwd_h2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GIMG));
    attr_set = true;
  }
  if (tk.dgh_n_only) hipLaunchKernelGGL(gru_scan_bwd_h2_kernel<true>, dim3(tk.Rm / 32), dim3(256), 2 * GIMG, s, tk);
  else hipLaunchKernelGGL(gru_scan_bwd_h2_kernel<false>, dim3(tk.Rm / 32), dim3(256), 2 * GIMG, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}
