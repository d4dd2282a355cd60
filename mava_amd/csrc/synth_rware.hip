// Synthetic RWARE-shaped environment step (SURVEY §8d "synthetic inputs"): the environment is
// NOT simulated (Jumanji's RobotWarehouse is third-party JAX code that is not available); this
// kernel produces observations / masks / rewards / dones with RWARE's shapes and statistics and
// reproduces the wrapper semantics the learner records:
//   * AgentIDWrapper (mava/wrappers/observation.py:41-53): one-hot agent id prepended to
//     agents_view only;
//   * RwareWrapper / get_global_state (mava/wrappers/jumanji.py:53-59,128-143): global_state =
//     concatenation of the RAW agent views (identical for every agent), team reward repeated per
//     agent, step_count per agent;
//   * AutoResetWrapper (mava/wrappers/auto_reset_wrapper.py:88-101): a terminal step returns the
//     reset observation with step_count 0;
//   * RecordEpisodeMetrics (mava/wrappers/episode_metrics.py:78-111): episode_return /
//     episode_length / is_terminal_step bookkeeping (return uses mean over agents of the reward).
// Distribution (seeded Philox4x32-10, counter = (entity, global step, word group, stream)):
//   agents_view = [one-hot id (A) | 2 grid coordinates U{0..9} | O-2 Bernoulli(51/256) bits]
//   action_mask = all legal except action 1 which is illegal w.p. 51/256
//   reward      = 1.0 w.p. 0.02 (team), done = step_count reaches time_limit or w.p. 0.002
//   reward_mode 1 ("match", opt-in, for learning tests): team reward = fraction of the env's agents whose action equals
//   (first grid coordinate of the observation they acted on) mod n_actions - the only action-dependent quantity
// Restated bit-for-bit in oracle/synth_env.py.
#include "common.h"

namespace {

constexpr uint32_t ENV_STREAM = 0x454E5653u;  // "ENVS"

struct SynthArgs {
  int E, A, O, nA;
  int gs_tiles;       // 1: global_state (E, A*O) shared by the agents; A: (E, A, A*O) tiled copy
  int S;              // 0: global_state = concatenated raw views (RWARE); > 0: an independent S-feature state
                      //    vector per env (SMAX-shaped: mava/wrappers/jaxmarl.py:326-373 world state)
  int time_limit;
  uint32_t seed_lo, seed_hi;
  uint32_t t;         // global step counter (unique per call)
  const uint32_t* t_base;  // optional device word added to t (captured HIP graphs replay with a moving counter)
  uint32_t env_offset;  // global id of env 0 (rank / replica offset)
  // state: step_count (E, A) - one private copy per agent thread (no cross-thread hazard); rest (E)
  int32_t* step_count;
  float* run_return;
  int32_t* run_length;
  float* ep_return;
  int32_t* ep_length;
  // outputs
  float* agents_view;   // (E, A, A+O)
  float* global_state;  // (E, gs_tiles, A*O)
  uint8_t* action_mask; // (E, A, nA)
  int32_t* obs_step_count;  // (E, A)
  float* reward;        // (E, A) or null (reset)
  uint8_t* done;        // (E, A) or null
  float* info_return;   // (E) or null
  int32_t* info_length; // (E) or null
  uint8_t* info_terminal;  // (E) or null
  int is_reset;
  const int32_t* action;   // (E, A) actions taken on the previous observation; read only when reward_mode == 1
  int reward_mode;         // 0: Bernoulli(0.02) team reward; 1: "match" (see the header)
};

// One thread per 16-feature CHUNK of a raw view (= one Philox block: every draw is a pure function of
// (entity, step, word group) on the counter-based RNG, so a thread owns 16 consecutive output floats and no thread
// needs another thread's draw), written with 8-byte stores where the row layout allows; the chunk-0 thread also
// writes the row's one-hot id and its two grid coordinates.  A thread per output float - the first version -
// recomputed the same Philox block 16 times and was VALU-bound at 13 us per step.  Then one thread per
// (env, agent) for the small per-agent / per-env data.
// The chunks of a wave leave as 16 store instructions, each writing 4 chunks x 16 consecutive floats (a full 64-byte run per
// 16 lanes): the lane that DREW a chunk stored it with 8 eight-byte stores whose 64 lanes were 64 bytes apart - every
// instruction touched 64 cache lines, and the kernel took 17 us per step at the SMAX shape for 10 MB.  The values cross the
// lanes through a wave-private LDS tile ([lane][17]: conflict-free both ways); n = 0 for a lane without a chunk.
__device__ __forceinline__ void wave_store_chunks(float* base, long off, int n, const float* tile, int lane) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int owner = 4 * k + (lane >> 4), e = lane & 15;
    const long o = __shfl(off, owner, 64);
    const int nn = __shfl(n, owner, 64);
    const float val = tile[owner * 17 + e];
    if (e < nn) base[o + e] = val;
  }
}

__global__ __launch_bounds__(256) void synth_rware_kernel(SynthArgs a) {
  __shared__ float xch[256 * 17];
  a.t += a.t_base ? *a.t_base : 0u;
  const uint32_t E = a.E, A = a.A, O = a.O, W = A + O;
  const uint32_t nch = max(1u, (O - 2 + 15) / 16);        // bit chunks per raw view (>= 1: chunk 0 also writes id + coordinates)
  const uint32_t n_view = E * A * nch;                    // (entity, chunk) threads
  const uint32_t gsw = a.S > 0 ? (uint32_t)a.S : A * O;   // width of one global-state row
  const uint32_t nch_s = a.S > 0 ? max(1u, ((uint32_t)a.S - 2 + 15) / 16) : 0;
  const uint32_t n_state = E * nch_s;                     // (env, chunk) threads of an independent state vector
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float* const tile = xch + (threadIdx.x >> 6) * 64 * 17;
  const bool is_chunk = gid < n_view + n_state;
  if (__ballot(is_chunk) != 0) {  // (wave-uniform: the chunk threads of a wave store together)
    const bool is_state = is_chunk && gid >= n_view;
    const uint32_t i = is_state ? gid - n_view : gid;
    const uint32_t per = is_state ? nch_s : nch;
    const uint32_t row = is_chunk ? i / per : 0u, c = is_chunk ? i - row * per : 0u;  // row = entity (env*A + agent) or env
    const uint32_t e = is_state ? row : row / A, ag = is_state ? 0u : row - e * A;
    const uint32_t ent = is_state ? (0x80000000u | (a.env_offset + e)) : ((a.env_offset + e) * A + ag);
    const uint32_t nf = is_state ? (uint32_t)a.S : O;     // features of this raw row
    const Philox4 r = philox4x32_10(ent, a.t, c, ENV_STREAM, a.seed_lo, a.seed_hi);
    const uint32_t wds[4] = {r.x, r.y, r.z, r.w};
    const uint32_t f0 = 2 + 16 * c;                       // first feature of the chunk
    const int n = is_chunk ? (int)min(16u, nf - f0) : 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[lane * 17 + q] = (((wds[q >> 2] >> (8 * (q & 3))) & 0xFFu) < 51u) ? 1.0f : 0.0f;
    float c0 = 0.0f, c1 = 0.0f;
    if (is_chunk && c == 0) {                              // the row's grid coordinates (features 0, 1)
      const Philox4 cm = philox4x32_10(ent, a.t, 0xFFFFu, ENV_STREAM, a.seed_lo, a.seed_hi);
      c0 = (float)(cm.x % 10u);
      c1 = (float)(cm.y % 10u);
    }
    const bool view = is_chunk && !is_state;
    // agents_view rows
    wave_store_chunks(a.agents_view, (long)row * W + A + f0, view ? n : 0, tile, lane);
    if (view && c == 0) {
      float* av = a.agents_view + (long)row * W;
      for (uint32_t k2 = 0; k2 < A; ++k2) av[k2] = (k2 == ag) ? 1.0f : 0.0f;
      av[A] = c0;
      av[A + 1] = c1;
    }
    // global state: the concatenated raw views (S == 0: written by the view threads) or an independent vector (state threads)
    const bool gs_on = is_chunk && ((a.S == 0) ? !is_state : is_state);
    for (int t2 = 0; t2 < a.gs_tiles; ++t2) {
      const long gbase = ((long)e * a.gs_tiles + t2) * gsw + (is_state ? 0L : (long)ag * O);
      wave_store_chunks(a.global_state, gbase + f0, gs_on ? n : 0, tile, lane);
      if (gs_on && c == 0) {
        a.global_state[gbase] = c0;
        a.global_state[gbase + 1] = c1;
      }
    }
  }
  if (is_chunk) return;
  const uint32_t k = gid - n_view - n_state;
  if (k >= E * A) return;
  const uint32_t e = k / A, ag = k - e * A;
  const uint32_t env_id = a.env_offset + e;

  // ---- per-env draws (recomputed by every agent thread of the env: identical values)
  Philox4 ev = philox4x32_10(env_id, a.t, 0u, ENV_STREAM ^ 1u, a.seed_lo, a.seed_hi);
  const int sc_old = a.is_reset ? 0 : a.step_count[k];
  float rew = (!a.is_reset && u01_open(ev.x) < 0.02f) ? 1.0f : 0.0f;
  if (a.reward_mode == 1 && !a.is_reset) {
    // the observation the agents acted on was generated at step t - 1: its first coordinate is recomputed, not re-read
    int hits = 0;
    for (uint32_t a2 = 0; a2 < A; ++a2) {
      const Philox4 pc = philox4x32_10(env_id * A + a2, a.t - 1u, 0xFFFFu, ENV_STREAM, a.seed_lo, a.seed_hi);
      hits += (a.action[e * A + a2] == (int32_t)((pc.x % 10u) % (uint32_t)a.nA)) ? 1 : 0;
    }
    rew = (float)hits / (float)A;
  }
  const int sc_new = sc_old + 1;
  const bool term = !a.is_reset && ((sc_new >= a.time_limit) || (u01_open(ev.y) < 0.002f));
  const int sc_obs = (a.is_reset || term) ? 0 : sc_new;

  Philox4 cm = philox4x32_10(env_id * A + ag, a.t, 0xFFFFu, ENV_STREAM, a.seed_lo, a.seed_hi);
  uint8_t* mk = a.action_mask + (long)k * a.nA;
  for (int i = 0; i < a.nA; ++i) mk[i] = 1;
  if (a.nA > 1 && ((cm.z & 0xFFu) < 51u)) mk[1] = 0;
  a.obs_step_count[k] = sc_obs;
  a.step_count[k] = sc_obs;

  if (!a.is_reset) {
    a.reward[k] = rew;
    a.done[k] = term ? 1 : 0;
  }
  // ---- per-env state and episode metrics (agent 0's thread)
  if (ag == 0) {
    if (a.is_reset) {
      a.run_return[e] = 0.0f;
      a.run_length[e] = 0;
      a.ep_return[e] = 0.0f;
      a.ep_length[e] = 0;
    } else {
      // episode_metrics.py:88-111 (mean over agents of a repeated team reward == the reward)
      const float new_ret = a.run_return[e] + rew;
      const int new_len = a.run_length[e] + 1;
      const float ret_info = term ? new_ret : a.ep_return[e];
      const int len_info = term ? new_len : a.ep_length[e];
      a.info_return[e] = ret_info;
      a.info_length[e] = len_info;
      a.info_terminal[e] = term ? 1 : 0;
      a.run_return[e] = term ? 0.0f : new_ret;
      a.run_length[e] = term ? 0 : new_len;
      a.ep_return[e] = ret_info;
      a.ep_length[e] = len_info;
    }
  }
}

}  // namespace

extern "C" int mava_synth_rware_step(int E, int A, int O, int n_actions, int gs_tiles, int state_dim, int time_limit,
                                     uint64_t seed, uint32_t t, const uint32_t* t_base, uint32_t env_offset, int is_reset,
                                     int32_t* step_count, float* run_return, int32_t* run_length,
                                     float* ep_return, int32_t* ep_length, float* agents_view,
                                     float* global_state, uint8_t* action_mask,
                                     int32_t* obs_step_count, float* reward, uint8_t* done,
                                     float* info_return, int32_t* info_length,
                                     uint8_t* info_terminal, const int32_t* action, int reward_mode,
                                     hipStream_t s) {
  MAVA_ARG_CHECK(E >= 0 && A >= 1 && O >= 2 && n_actions >= 1 && time_limit >= 1, 0,
                 "mava_synth_rware_step: bad shape E=%d A=%d O=%d nA=%d", E, A, O, n_actions);
  MAVA_ARG_CHECK((gs_tiles == 1 || gs_tiles == A) && state_dim >= 0, 1,
                 "mava_synth_rware_step: gs_tiles must be 1 or A, state_dim >= 0");
  if (E == 0) return MAVA_OK;
  MAVA_ARG_CHECK(step_count && run_return && run_length && ep_return && ep_length && agents_view &&
                     global_state && action_mask && obs_step_count,
                 2, "mava_synth_rware_step: null state/observation pointer");
  MAVA_ARG_CHECK(is_reset || (reward && done && info_return && info_length && info_terminal), 3,
                 "mava_synth_rware_step: null transition pointer");
  SynthArgs a;
  a.E = E; a.A = A; a.O = O; a.nA = n_actions; a.gs_tiles = gs_tiles; a.S = state_dim; a.time_limit = time_limit;
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.t = t; a.t_base = t_base; a.env_offset = env_offset;
  a.step_count = step_count; a.run_return = run_return; a.run_length = run_length;
  a.ep_return = ep_return; a.ep_length = ep_length; a.agents_view = agents_view;
  a.global_state = global_state; a.action_mask = action_mask; a.obs_step_count = obs_step_count;
  a.reward = reward; a.done = done; a.info_return = info_return; a.info_length = info_length;
  a.info_terminal = info_terminal; a.is_reset = is_reset;
  MAVA_ARG_CHECK(reward_mode == 0 || (reward_mode == 1 && (is_reset || action != nullptr)), 5,
                 "mava_synth_rware_step: reward_mode %d needs the (E, A) action array", reward_mode);
  a.action = action; a.reward_mode = reward_mode;
  MAVA_ARG_CHECK(state_dim == 0 || state_dim >= 2, 1, "mava_synth_rware_step: state_dim must be 0 or >= 2");
  const long nch = (O - 2 + 15) / 16 > 0 ? (O - 2 + 15) / 16 : 1;
  const long nch_s = state_dim > 0 ? ((state_dim - 2 + 15) / 16 > 0 ? (state_dim - 2 + 15) / 16 : 1) : 0;
  const long total = (long)E * A * nch + (long)E * nch_s + (long)E * A;
  MAVA_ARG_CHECK(total < (1L << 32) && (long)E * A * (A + O) < (1L << 31), 4,
                 "mava_synth_rware_step: %ld threads exceed 32-bit indexing", total);
  hipLaunchKernelGGL(synth_rware_kernel, dim3(mava_cdiv(total, 256)), dim3(256), 0, s, a);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}
