// Task descriptions shared by the recurrent path's dense / X^T Y kernels (rec_dense.hip: exact-f32 MFMAs;
// rec_dense_h2.hip: split-f16 operands).
#pragma once
#include "common.h"

namespace mava_rec {

struct DenseTask {
  const float* x;        // T32 (rows x K) or row-major gather source
  int x_rowmajor;        // 1: x is row-major (rows_x x K); row r of the batch reads x[xrow(r)]
  // row-major gather (time-major trajectory): batch row q -> t = q / Rm, m = q % Rm, env = idx[m / A],
  // agent = m % A, source row = ((t * E + env) * A + agent) / xshare
  const int32_t* idx;    // (Rm / A) env ids of the minibatch, or null (identity)
  int Rm, E, A, xshare;
  int x_ld;              // row stride (floats) of the row-major source (>= K; lets a call read a column block)
  int accumulate;        // 1: start from the existing y (T32) instead of the bias (K-chunked products)
  const float* w;        // (K x N) row-major, row stride ldw
  int ldw;
  const float* bias;     // (N) or null
  const float* gate;     // T32 (rows x N) or null: output multiplied by (gate > 0)
  float* y;              // T32 (rows x N)
  int K, N, rows, relu;
  int y_ld;              // features per y / gate tile (>= N: a call can write a column block of a wider matrix)
};

__device__ __forceinline__ long gather_row(const DenseTask& tk, int q) {
  const int t = q / tk.Rm, m = q - t * tk.Rm;
  const int e_local = m / tk.A, a = m - e_local * tk.A;
  const int env = tk.idx ? tk.idx[e_local] : e_local;
  return ((long)((long)t * tk.E + env) * tk.A + a) / tk.xshare;
}

struct XtyTask {
  const float* x;      // T32 (rows x K) or row-major gather source
  int x_rowmajor;
  const int32_t* idx;
  int Rm, E, A, xshare;
  int x_ld;
  const float* y;      // T32 (rows x N)
  int K, N, rows;
  int y_ld;            // features per y tile (>= N: a call can read a column block of a wider matrix)
  float* slab;         // (gridDim.x, slab_stride): [dW (K x N row-major) | db (N)]
  long slab_stride;
  int want_bias;
  float out_scale;     // dW and db are multiplied by this at the slab write (the backward chain runs in scaled units)
  const float* y_tail; // null, or T32 (rows x y_tail_ld): features [y_split, N) of Y are its features [0, N - y_split)
  int y_split;         // (a multiple of 32: the BPTT scan's dgh = [dgi's r and z thirds | its own n third])
  int y_tail_ld;
};


struct ScanTask {
  int T, Rm, E, A;           // Rm sequences (multiple of 32); E envs in the external arrays
  const int32_t* idx;        // (Rm / A) env ids or null
  const uint8_t* done;       // external (T, E, A) u8: flag ENTERING each step
  const float* h0;           // initial hidden state: T32 (Rm x 128) if h0_t32 else external (E, A, 128)
  int h0_t32;
  const float* wh;           // (128 x 384) row-major [hr | hz | hn]
  const float* bhn;          // (128)
  // forward
  const float* gi;           // T32 (T*Rm x 384)
  float* hs;                 // T32 (T*Rm x 128) h after each step
  float* hprev;              // T32 (T*Rm x 128) masked h entering each step (null: not stored)
  float* saved;              // T32 (T*Rm x 512) [r | z | n | hn_lin]     (null: not stored)
  // backward
  const float* dh_out;       // T32 (T*Rm x 128)
  float* dgi;                // T32 (T*Rm x 384)
  float* dgh;                // T32 (T*Rm x 384), or (T*Rm x 128) = its n third alone when dgh_n_only (r and z thirds == dgi's)
  int dgh_n_only;
};

__device__ __forceinline__ long ext_row(const ScanTask& tk, int t, int m) {
  const int e_local = m / tk.A, a = m - e_local * tk.A;
  const int env = tk.idx ? tk.idx[e_local] : e_local;
  return ((long)t * tk.E + env) * tk.A + a;
}

}  // namespace mava_rec
using mava_rec::DenseTask;
using mava_rec::ScanTask;
using mava_rec::ext_row;
using mava_rec::XtyTask;
using mava_rec::gather_row;

// f16x2 forms (rec_dense_h2.hip); T32 inputs only.  Return 1 when the shape is not instantiated (caller runs the f32 kernel).
int mava_rec_dense_h2_launch(const DenseTask& tk, hipStream_t s);
int mava_rec_xty_h2_launch(const XtyTask& tk, int n_slab, hipStream_t s);
// rec_gru_h2.hip: the GRU scans with the recurrent products on split-f16 operands
int mava_gru_scan_fwd_h2_launch(const ScanTask& tk, hipStream_t s);
int mava_gru_scan_bwd_h2_launch(const ScanTask& tk, hipStream_t s);
