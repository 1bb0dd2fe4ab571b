/* swarm_policy.h -- C ABI of the fused policy MLP used by the device-resident rollout.
 *
 * Replaces, for inference during rollouts, the reference's actor forward
 *   /root/reference/marl_llm/algorithm/utils/networks.py:6-44   (MLPNetwork: fc1..fc4, leaky_relu x3, tanh)
 * called from /root/reference/marl_llm/algorithm/utils/agents.py:69-96 (DDPGAgent.step) on torch.Tensor(obs).
 * One HIP kernel (bf16 MFMA, fp32 accumulate) maps the env's observation rows [rows][in_dim] (fp32, device) to actions
 * [rows][act_dim] (fp32, device); weights are given once, in torch.nn.Linear layout ([out][in] row-major, fp32, host).
 * Same numerical contract as torch.autocast(bfloat16) on that module.  No CPU path: without a HIP device create fails.
 */
#ifndef SWARM_POLICY_H
#define SWARM_POLICY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWARM_POLICY_OK 0
#define SWARM_POLICY_ERR_INVALID 1
#define SWARM_POLICY_ERR_HIP 2

typedef struct swarm_policy swarm_policy_t;

/* w1 [hidden][in_dim], w2 / w3 [hidden][hidden], w4 [act_dim][hidden], b* the biases; all fp32 HOST pointers.
 * Supported: in_dim <= 192 (multiple of 4), hidden <= 192, act_dim <= 4.  device < 0: the current device. */
int  swarm_policy_create(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                         const float *w4, const float *b4, int in_dim, int hidden, int act_dim, int device, swarm_policy_t **out);
void swarm_policy_destroy(swarm_policy_t *p);

/* act[rows][act_dim] = tanh(fc4(lrelu(fc3(lrelu(fc2(lrelu(fc1(obs[rows][in_dim])))))))); obs / act are DEVICE pointers, rows
 * densely packed; stream is a hipStream_t (NULL: the default stream).  Asynchronous. */
int  swarm_policy_forward(swarm_policy_t *p, const float *obs, int64_t rows, float *act, void *stream);

/* The same with bfloat16 observation rows (the env's SWARM_BF16 output, swarm_env.h): each 16-byte load is one MFMA
 * operand fragment, no conversion.  in_dim must be a multiple of 8 (16-byte aligned rows).  Identical results to
 * swarm_policy_forward on the same values held in float32. */
int  swarm_policy_forward_bf16(swarm_policy_t *p, const void *obs_bf16, int64_t rows, float *act, void *stream);

/* The rollout's exploring actor in one launch (agents.py:82-96, continuous branch): act = clamp(actor(obs) + noise_scale *
 * N(0, 1), -1, 1).  The normals come from a counter-based generator keyed by (seed, step, row, component) -- any row range
 * reproducible on any rank, nothing to store -- evaluated in the kernel's epilogue; noise_scale <= 0: plain forward.
 * `act` may point anywhere on the device (e.g. straight into a replay-buffer slot).  obs_is_bf16 as the two calls above. */
int  swarm_policy_forward_explore(swarm_policy_t *p, const void *obs, int obs_is_bf16, int64_t rows, float *act,
                                  float noise_scale, uint64_t seed, uint64_t step, void *stream);

/* Arithmetic of the forward calls.  SWARM_POLICY_BF16 (default): operands rounded to bfloat16, fp32 sums -- the contract of
 * torch.autocast(bfloat16), ~4e-2 from the reference's fp32 actor on actions in [-1, 1].  SWARM_POLICY_BF16X3: operands split
 * into a high and a low bfloat16 part, three MFMAs per product (hi hi + hi lo + lo hi), fp32 sums -- within ~1e-4 of the fp32
 * actor (networks.py:6-44), about 1.9x the time. */
#define SWARM_POLICY_BF16   0
#define SWARM_POLICY_BF16X3 1
int  swarm_policy_set_precision(swarm_policy_t *p, int precision);

const char *swarm_policy_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
