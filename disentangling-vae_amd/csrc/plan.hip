// dvae_plan_run: a recorded launch list of C-ABI calls replayed by ONE foreign call.
//
// A training iteration at a fixed batch size is the same ~60 entry-point calls with the same arguments (device pointers,
// sizes, streams) every step; below a few hundred images per GPU the host cannot issue them through ctypes as fast as the
// GPU retires them on a slow box (~2.5 us per foreign call, 0.15 ms per step: profiles/r03_final_sweep.txt).  The host side
// records the calls once (disvae_amd/_lib.py), packs every argument into 64 bits, and hands the array back here each step.
// An entry names its entry point by index into OPS (dvae_plan_op(name)); the trampolines below are generated from the
// DECLARED signatures of include/dvae_hip.h, so every argument is converted to exactly the type its entry point takes.
#include <string.h>
#include <type_traits>
#include <utility>
#include "common.h"

namespace dvae {

template <typename T>
static inline T plan_arg(uint64_t v) {
  if constexpr (std::is_pointer_v<T>) {
    return reinterpret_cast<T>(v);
  } else if constexpr (std::is_floating_point_v<T>) {
    float f;
    const uint32_t u = (uint32_t)v;                  // fp32 bit pattern in the low word
    memcpy(&f, &u, sizeof(f));
    return (T)f;
  } else {
    return static_cast<T>((int64_t)v);
  }
}
template <typename... A, size_t... I>
static inline int plan_invoke(int (*fn)(A...), const uint64_t* a, std::index_sequence<I...>) {
  return fn(plan_arg<A>(a[I])...);
}
template <typename... A>
static inline int plan_call(int (*fn)(A...), const uint64_t* a, int nargs) {
  static_assert(sizeof...(A) <= DVAE_PLAN_MAX_ARGS, "raise DVAE_PLAN_MAX_ARGS");
  if (nargs != (int)sizeof...(A)) return -100;
  return plan_invoke(fn, a, std::index_sequence_for<A...>{});
}

struct PlanOp {
  const char* name;
  int (*run)(const uint64_t*, int);
};
#define DVAE_OP(f) {#f, [](const uint64_t* a, int n) { return plan_call(f, a, n); }}
static const PlanOp OPS[] = {
    DVAE_OP(dvae_conv4s2_fwd), DVAE_OP(dvae_conv4s2_dgrad), DVAE_OP(dvae_conv4s2_wgrad), DVAE_OP(dvae_convT4s2_fwd),
    DVAE_OP(dvae_convT4s2_dgrad), DVAE_OP(dvae_convT4s2_wgrad), DVAE_OP(dvae_convT4s2_sigmoid_recon_fwd),
    DVAE_OP(dvae_stage_weights), DVAE_OP(dvae_convT3_fwd_staged), DVAE_OP(dvae_conv32_down), DVAE_OP(dvae_conv32_up),
    DVAE_OP(dvae_conv1_fwd_bits), DVAE_OP(dvae_conv32_up_bits), DVAE_OP(dvae_convT3_dgrad_bits),
    DVAE_OP(dvae_u8_to_f32), DVAE_OP(dvae_conv4s2_fwd_u8), DVAE_OP(dvae_conv4s2_wgrad_u8),
    DVAE_OP(dvae_convT4s2_sigmoid_recon_fwd_u8), DVAE_OP(dvae_relayout), DVAE_OP(dvae_linear_fwd), DVAE_OP(dvae_linear_dgrad),
    DVAE_OP(dvae_linear_wgrad), DVAE_OP(dvae_linear_wgrad_grouped), DVAE_OP(dvae_fc_chain_fwd), DVAE_OP(dvae_fc_chain_bwd),
    DVAE_OP(dvae_reparam_kl_fwd), DVAE_OP(dvae_kl_finish), DVAE_OP(dvae_reparam_kl_bwd), DVAE_OP(dvae_kl_normal_bwd),
    DVAE_OP(dvae_reduce_sum), DVAE_OP(dvae_recon_loss), DVAE_OP(dvae_sigmoid_bwd), DVAE_OP(dvae_btcvae_fwd),
    DVAE_OP(dvae_btcvae_bwd), DVAE_OP(dvae_permute_dims), DVAE_OP(dvae_disc_losses), DVAE_OP(dvae_latent_entropy),
    DVAE_OP(dvae_loss_pack), DVAE_OP(dvae_loss_finalize), DVAE_OP(dvae_loss_epilogue), DVAE_OP(dvae_set_coef), DVAE_OP(dvae_add),
    DVAE_OP(dvae_axpby), DVAE_OP(dvae_swap_outer),
    DVAE_OP(dvae_stream_order), DVAE_OP(dvae_event_record), DVAE_OP(dvae_event_wait), DVAE_OP(dvae_comm_allreduce), DVAE_OP(dvae_comm_allgather), DVAE_OP(dvae_comm_reducescatter),
    DVAE_OP(dvae_comm_broadcast), DVAE_OP(dvae_comm_group_start), DVAE_OP(dvae_comm_group_end),
};
#undef DVAE_OP
constexpr int N_OPS = (int)(sizeof(OPS) / sizeof(OPS[0]));

}  // namespace dvae

extern "C" {

int dvae_plan_op(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < dvae::N_OPS; ++i)
    if (strcmp(dvae::OPS[i].name, name) == 0) return i;
  return -1;
}

int dvae_plan_run(const dvae_plan_entry* entries, int n) {
  if (!entries || n < 0) {
    dvae::set_error("dvae_plan_run: invalid argument");
    return -1;
  }
  for (int i = 0; i < n; ++i) {
    const dvae_plan_entry& e = entries[i];
    if (e.op < 0 || e.op >= dvae::N_OPS) {
      dvae::set_error("dvae_plan_run: entry %d: unknown op %d", i, e.op);
      return -1;
    }
    const int rc = dvae::OPS[e.op].run(e.args, e.nargs);
    if (rc == -100) {
      dvae::set_error("dvae_plan_run: entry %d (%s): %d arguments recorded", i, dvae::OPS[e.op].name, e.nargs);
      return -1;
    }
    if (rc != 0) return rc;                            // the entry point has set the error text
  }
  return 0;
}

}  // extern "C"
