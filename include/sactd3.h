/*
 * sactd3.h -- C ABI of the MI355X-native SAC/TD3 update engine (libsactd3_hip.so).
 *
 * The reference (lionelblonde/sac-td3-cudagraphs-pytorch) has no FFI of its own: its boundary is
 * the duck-typed Python `Agent` object that orchestrator.py drives.  Every
 * entry point below therefore cites the reference *call site / method* it stands in for; the
 * Python mirror of that object lives in sac-td3-cudagraphs-pytorch_amd/agent.py and binds these
 * symbols with ctypes.
 *
 * Conventions
 *   - plain C types only; all host pointers are caller-owned and only read/written during the call;
 *   - every function returns 0 on success, a negative SACTD3_E* code on failure and never throws;
 *     `sactd3_last_error` gives the text of the last failure on that engine (NULL engine: of the
 *     last failed `sactd3_create` on this thread);
 *   - calls are asynchronous on the engine's HIP stream unless marked [sync];
 *   - one engine = one learner = one GPU; an engine is not thread-safe, different engines are.
 *   - all arithmetic is fp32, hidden width is 256 (agents/agent.py:56,101 hard-codes (256,256)).
 *   - the library reads NO environment variable: every behaviour is a field of sactd3_config.  (Kernel-selection A/B switches
 *     exist only in the separate tuning build, `make -C csrc tune` -> libsactd3_hip_tune.so, -DSACTD3_TUNING; they are listed in
 *     csrc/engine.hip:create_impl and never compiled into libsactd3_hip.so -- tests/test_abi.py checks the shipped binary.)
 */
#ifndef SACTD3_H
#define SACTD3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SACTD3_ABI_VERSION 1

enum {
  SACTD3_OK = 0,
  SACTD3_EINVAL = -1,   /* bad argument / configuration                      */
  SACTD3_EHIP = -2,     /* a HIP runtime call failed (text in last_error)    */
  SACTD3_ESTATE = -3,   /* call not valid in the current state (e.g. empty buffer) */
  SACTD3_ENODEV = -4    /* no usable gfx950 device                            */
};

/* Which parameter set (`which` arguments).  Layout of the float arrays exchanged with the host is the
 * reference's state_dict order (agents/nets.py:66-82, verified key list in SURVEY.md 8-A12):
 *   fc_block_1.fc.weight [256,in] | .bias [256] | (ln.weight [256] | ln.bias [256])   -- LN pair only if layer_norm
 *   fc_block_2.fc.weight [256,256] | .bias | (ln.weight | ln.bias) | head.weight [nh,256] | head.bias [nh]
 * all row-major, unpadded; the twin critics are two such blocks back to back (= the dense-stacked
 * [2, ...] tensors of agents/agent.py:106 viewed net-major). */
enum {
  SACTD3_ACTOR = 0,
  SACTD3_CRITICS = 1,
  SACTD3_ACTOR_TARGET = 2,
  SACTD3_CRITICS_TARGET = 3,
  SACTD3_LOG_ALPHA = 4
};

/* Noise sites (sactd3_set_noise).  Draw order per iteration is the reference's:
 * critic draw -> [actor draw -> alpha draw] x actor_update_delay (agents/agent.py:205,254,298). */
enum {
  SACTD3_SITE_CRITIC = 0,  /* SAC: eps of a'~pi(s') (agent.py:205); TD3: smoothing noise N(0,1) (agent.py:197) */
  SACTD3_SITE_ACTOR0 = 1,  /* eps of a~pi(s), 1st actor update of the iteration (agent.py:254) */
  SACTD3_SITE_ACTOR1 = 2,  /*                 2nd actor update (fused sactd3_step only)       */
  SACTD3_SITE_ALPHA0 = 3,  /* eps of the fresh alpha-loss draw (agent.py:298)                 */
  SACTD3_SITE_ALPHA1 = 4,
  SACTD3_SITE_PREDICT = 5, /* exploration draw in predict (nets.py:156-158,225)               */
  SACTD3_NUM_SITES = 6
};

/* Metrics slots (sactd3_read_metrics): the keys the reference's update methods return
 * (agents/agent.py:238-242,305-311). */
enum {
  SACTD3_M_QF_LOSS = 0,
  SACTD3_M_ACTOR_LOSS = 1,
  SACTD3_M_ALPHA_LOSS = 2,
  SACTD3_M_ALPHA = 3,
  SACTD3_NUM_METRICS = 8
};

/* Numeric hyper-parameters = the keys of tasks/defaults/{sac,td3}.yml the hot path reads
 * (agents/agent.py:47-58,115-139,194-228,284-331; orchestrator.py:345-348). */
typedef struct sactd3_config {
  int32_t abi_version;          /* must be SACTD3_ABI_VERSION */
  int32_t ob_dim;               /* net_shapes["ob_shape"][-1] */
  int32_t ac_dim;               /* net_shapes["ac_shape"][-1] */
  int32_t batch_size;           /* sac.yml:38 */
  int32_t rb_capacity;          /* sac.yml:40 (rows) */
  int32_t max_envs;             /* largest n accepted by predict / rb_extend in one call (>= num_envs) */
  int32_t prefer_td3_over_sac;  /* sac.yml:42 */
  int32_t layer_norm;           /* sac.yml:29 */
  int32_t autotune;             /* sac.yml:47 */
  int32_t bcq_style_targ_mix;   /* td3.yml:43 */
  int32_t targ_actor_smoothing; /* td3.yml:46 */
  int32_t actor_update_delay;   /* sac.yml:44 */
  int32_t crit_targ_update_freq;/* sac.yml:45 (ignored for TD3, agent.py:323) */
  int32_t use_graphs;           /* 1: hipGraph replay (the reference's cudagraphs: true); 0: eager launches */
  int32_t device_id;            /* HIP device ordinal */
  int32_t reserved0;
  float actor_lr, qnets_lr, log_alpha_lr;   /* sac.yml:32-33,48 */
  float gamma, polyak, alpha_init;          /* sac.yml:39,41,46 */
  float clip_norm;                          /* sac.yml:34; <= 0 disables (agent.py:284) */
  float td3_std, td3_c, actor_noise_std;    /* td3.yml:45-48 */
  float adam_beta1, adam_beta2, adam_eps;   /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 (agent.py:115-139) */
  float reserved1;
  uint64_t seed;                /* keys the engine's Philox4x32-10 streams (main.py:145-146 seeds torch instead) */
} sactd3_config;

typedef struct sactd3_engine sactd3_engine;

/* Fill `cfg` with tasks/defaults/sac.yml (td3 == 0) or td3.yml (td3 != 0) values; dims/seed left 0. */
void sactd3_default_config(sactd3_config* cfg, int td3);

/* Agent.__init__ (agents/agent.py:24-144).  min_ac/max_ac: [ac_dim] action bounds (agent.py:35-36).
 * Parameters start at zero weights / LN gamma=1: the caller supplies the reference's orthogonal
 * init through sactd3_set_params (the Python mirror does, with torch's own orthogonal_). */
int sactd3_create(const sactd3_config* cfg, const float* min_ac, const float* max_ac, sactd3_engine** out);
void sactd3_destroy(sactd3_engine* e);
const char* sactd3_last_error(const sactd3_engine* e);
int sactd3_abi_version(void);

/* ---- parameters / optimiser state: Agent.save / load_from_disk (agents/agent.py:333-371) ---- */
int64_t sactd3_param_count(const sactd3_engine* e, int which);            /* floats exchanged for `which` */
int sactd3_get_params(sactd3_engine* e, int which, float* dst);           /* [sync] */
int sactd3_set_params(sactd3_engine* e, int which, const float* src);     /* [sync] */
/* Adam state of the optimiser that owns `which` (ACTOR -> actor_optimizer, CRITICS -> q_optimizer,
 * LOG_ALPHA -> alpha_optimizer): exp_avg / exp_avg_sq in the same layout as the params, step count. */
int sactd3_get_adam_state(sactd3_engine* e, int which, float* exp_avg, float* exp_avg_sq, int64_t* step); /* [sync] */
int sactd3_set_adam_state(sactd3_engine* e, int which, const float* exp_avg, const float* exp_avg_sq, int64_t step); /* [sync] */

/* ---- replay buffer: TensorDictReplayBuffer(LazyTensorStorage(capacity, device)) (main.py:167-171) ---- */
/* rb.extend (orchestrator.py:100-113): n rows, row-major host arrays; `dones` is the reference's
 * `terminations` (== `dones`, orchestrator.py:107-108) as bytes. Rows go to the ring cursor, wrapping. */
int sactd3_rb_extend(sactd3_engine* e, const float* obs, const float* actions, const float* rewards,
                     const float* next_obs, const uint8_t* dones, int n);
int64_t sactd3_rb_len(const sactd3_engine* e);                             /* len(rb) (orchestrator.py:385) */
/* layout of one ring record, in floats: out = {record length, offset of s' (= padded width of [s | a]), padded width of s',
 * rb_capacity}; record = [s (ob_dim) | a (ac_dim) | 0-pad][s' (ob_dim) | 0-pad][r, d (0/1)][0-pad to a 64-byte multiple] */
int sactd3_rb_layout(const sactd3_engine* e, int32_t out[4]);
/* rb.extend (orchestrator.py:100-113) with n records ALREADY PACKED (sactd3_rb_layout) IN DEVICE MEMORY: the shared-replay
 * variant of BASELINE.json's north_star (not in the reference) all-gathers every rank's new rows over RCCL into one device
 * slab and appends it here with one kernel, no host round trip.  `records` must stay valid until sactd3_sync. */
int sactd3_rb_extend_device(sactd3_engine* e, const float* records, int n);
/* rb.sample(batch_size) (orchestrator.py:338): uniform-with-replacement indices from the engine's
 * Philox stream + gather into the engine-owned batch slot. */
int sactd3_rb_sample(sactd3_engine* e);
/* same gather with caller-chosen indices (parity tests: the reference's torch RNG stream is not reproduced) */
int sactd3_rb_sample_with_indices(sactd3_engine* e, const int64_t* idx, int n);
/* a caller-owned batch (what update_qnets(batch) receives in the reference), copied into the batch slot */
int sactd3_load_batch(sactd3_engine* e, const float* obs, const float* actions, const float* rewards,
                      const float* next_obs, const uint8_t* dones, int n);
/* read the batch slot back (any pointer may be NULL) [sync] */
int sactd3_read_batch(sactd3_engine* e, float* obs, float* actions, float* rewards, float* next_obs,
                      uint8_t* dones, int64_t* idx);
/* device-side synthetic fill of rows [0,n) for benchmarks: s,s',r ~ N(0,1), a ~ U(min,max), d ~ Bern(0.01) */
int sactd3_rb_fill_synthetic(sactd3_engine* e, int64_t n, uint64_t seed);

/* ---- noise injection (parity): eps [n, ac_dim] standard normals for `site`; sticky until cleared ---- */
int sactd3_set_noise(sactd3_engine* e, int site, const float* eps, int n);
int sactd3_clear_noise(sactd3_engine* e, int site);   /* site < 0: all sites; back to the native Philox draws */
int sactd3_read_noise(sactd3_engine* e, int site, float* eps, int n);      /* last used draws [sync] */

/* ---- the update path ---- */
int sactd3_update_qnets(sactd3_engine* e);    /* Agent.update_qnets  (agents/agent.py:183-242) on the batch slot */
int sactd3_update_actor(sactd3_engine* e);    /* Agent.update_actor  (agents/agent.py:244-318) on the batch slot */
/* Agent.update_targ_nets (agents/agent.py:320-331); the caller passes its already-incremented counter
 * exactly as the reference reads self.qnet_updates_so_far (orchestrator.py:342,352) */
int sactd3_update_targ_nets(sactd3_engine* e, int64_t qnet_updates_so_far);
/* One whole loop iteration of orchestrator.py:337-352 as ONE graph launch: sample+gather, critic
 * update, (if do_actor) actor_update_delay x actor(+alpha) updates on the same batch, Polyak
 * (subject to crit_targ_update_freq and the engine's own update counter). */
int sactd3_step(sactd3_engine* e, int do_actor);
/* actor_update_delay + 1 consecutive iterations of orchestrator.py:337-352 -- the first with the actor updates, the others
 * critic-only: one period of the schedule of :345-349 -- as ONE graph launch.  Equal to that many sactd3_step calls, bit for
 * bit.  Needs TD3 or crit_targ_update_freq == 1 (else SACTD3_ESTATE: issue the iterations with sactd3_step). */
int sactd3_step_period(sactd3_engine* e);
/* The first m iterations of a period (1 <= m <= actor_update_delay: the one with the actor updates + m - 1 critic-only ones) as ONE
 * graph launch -- what a run of iterations leaves behind its last whole period (orchestrator.py:337-352 with a number of
 * iterations that is not a multiple of the period).  Equal to sactd3_step(e, 1) followed by m - 1 sactd3_step(e, 0), bit for bit.
 * Same preconditions as sactd3_step_period. */
int sactd3_step_prefix(sactd3_engine* e, int m);
/* Capture and instantiate the hipGraphs of sactd3_step / sactd3_step_period now rather than at their first use (the reference's
 * CudaGraphModule captures after a warm-up inside the loop, orchestrator.py:313-315); nothing is launched, no state changes. */
int sactd3_instantiate_graphs(sactd3_engine* e);
/* Agent.predict (agents/agent.py:172-181): obs [n, ob_dim] host -> actions [n, ac_dim] host.  Stream-ordered behind whatever
 * update was issued before it (it acts with the updated parameters, as the reference does) and returns when ITS kernels have
 * finished: [sync] in that sense -- with at most 4 rows (16 for wide heads) the wait is a spin on a pinned host word the last
 * kernel publishes, otherwise a stream synchronisation. */
int sactd3_predict(sactd3_engine* e, const float* obs, int n, int explore, float* actions);

int sactd3_read_metrics(sactd3_engine* e, float out[SACTD3_NUM_METRICS]);  /* [sync] */
/* The engine's HIP stream (hipStream_t) and the DEVICE address of the metrics slots, for callers that want the values the
 * way the reference hands them out -- 0-dim device tensors that are only materialised at evaluation time (agents/agent.py:
 * 238-242,305-311; orchestrator.py:341,348,383): wrap `metrics + SACTD3_M_*` as a tensor and order the consumer's stream
 * after `stream` (the Python mirror does both with torch).  The memory is owned by the engine and overwritten by later updates. */
int sactd3_device_handles(sactd3_engine* e, void** stream, float** metrics);
int sactd3_sync(sactd3_engine* e);                                          /* [sync] */

/* ---- introspection for tests / profiling (not part of the reference surface) ---- */
/* copy a named internal device buffer to the host; returns the number of floats it holds (or < 0).
 * With dst == NULL only the size is returned. Names: see sactd3_debug_names(). [sync] */
int64_t sactd3_debug_read(sactd3_engine* e, const char* name, float* dst, int64_t max_floats);
const char* sactd3_debug_names(void);
/* number of kernel nodes in the instantiated graph of: 0 update_qnets, 1 update_actor, 2 step(do_actor=0), 3 step(do_actor=1), 4 step_period,
 * 5 the opening graph of a period that cannot use a precomputed opening pair, 6 / 7 step_prefix(1) / step_prefix(2) */
int sactd3_graph_kernel_count(sactd3_engine* e, int which_graph);
/* average device time in microseconds of `iters` back-to-back launches of one kernel of the path,
 * measured with hipEvents on the engine's stream: "gather" (a fresh index draw per launch), "polyak", "trunk_critics" (the 4-net
 * hidden-layer launch of update_qnets; on wide inputs it is two launches). [sync] */
int sactd3_time_kernel(sactd3_engine* e, const char* kernel, int iters, float* usec);
/* Per-node device time of one fused iteration (sactd3_step with this do_actor; do_actor == 2: of one whole period as
 * sactd3_step_period captures it): every kernel launch of the sequence
 * alone, `iters` times back to back between two HIP events on the engine's stream.  Returns the node count n (<= max_nodes)
 * and fills usec[n], flops[n] (2 x MACs of the GEMMs in the launch), bytes[n] (operands + results, each once),
 * threads[n] (grid x block, rocprofv3's Grid_Size); `names` receives n newline-terminated "kernel-instance:role" strings.  Consumes the learner's state (optimiser steps repeat on
 * stale gradients): call it on a scratch engine.  Stands in for nothing in the reference: SURVEY.md 8d measurement. [sync] */
int sactd3_time_nodes(sactd3_engine* e, int do_actor, int iters, int max_nodes, char* names, int names_cap,
                      float* usec, double* flops, double* bytes, int64_t* threads);
/* run the replay gather at an arbitrary batch size (own scratch outputs, rows of the engine's ring): the bandwidth sweep of SURVEY.md 8d [sync] */
int sactd3_time_gather_sweep(sactd3_engine* e, int batch, int iters, float* usec, double* algo_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SACTD3_H */
