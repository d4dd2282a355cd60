// Argument blocks of the fused recurrent acting step, shared by the exact-f32 kernel (rec_step.hip) and the split-f16
// kernel with pre-packed weights (rec_step_h2.hip).
#pragma once
#include "mlp_core.h"

struct RecNet {
  const float* params;   // [Wpre (din,128) | bpre | Wi (128,384) | bi | Wh (128,384) | bhn | Wpost | bpost | Whead (128,no) | bhead]
  const float* x;        // (rows_x, din) row-major; row r reads x[r / xshare]
  const uint8_t* done;   // flag ENTERING this step (resets the hidden state): row r reads done[r * done_stride]
  int done_stride;
  const float* h_in;     // T32 (rows x 128)
  float* h_out;          // T32 (rows x 128)
  int din, no, xshare, rows;
  // LDS carve (floats)
  int xs, ldx, nb1, et, ht, h2t, pt, w3, yp, end;
};

struct RecStepOut {
  const uint8_t* mask;   // (rows, no) or null
  uint32_t seed_lo, seed_hi, step, row_offset;
  int greedy;
  int32_t* action;
  float* log_prob;
  float* value;          // (rows_c * vbroadcast)
  int vbroadcast;
  float* action_f;       // continuous head (tanh_normal.h) when not null: (rows, no) actions; log_std follows bhead
  float min_scale;       // continuous head: scale = softplus(log_std) + min_scale
};


// rec_step_h2.hip.  Returns MAVA_OK, 1 when the shape is not instantiated (the caller runs the f32 kernel), or an error.
int mava_rec_step_h2_launch(const RecNet& actor, const RecNet& critic, const void* pack_a, const void* pack_c,
                            const RecStepOut& out, hipStream_t s);
