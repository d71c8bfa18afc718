/* A C consumer of include/rmi_hip.h: prints the size of every structure that crosses the boundary and the offset of every
 * field, as "struct.field offset size" lines.  tests/test_abi_cpu.py compiles it with gcc and compares the lines with the
 * ctypes mirrors of rmi_amd/_lib.py -- the only guard between an edit of the header and a silently mis-laid-out structure on
 * the other side of the FFI (the #[repr(C)] structs of INTEGRATION.md section 2 follow the same header). */
#include <stddef.h>
#include <stdio.h>

#include "rmi_hip.h"

#define F(S, f) printf(#S "." #f " %zu %zu\n", offsetof(S, f), sizeof(((S*)0)->f))
#define SZ(S) printf(#S " size %zu\n", sizeof(S))

int main(void) {
  printf("abi %d\n", RMI_HIP_ABI_VERSION);
  SZ(rmi_hip_model_params);
  F(rmi_hip_model_params, kind); F(rmi_hip_model_params, p); F(rmi_hip_model_params, ip);
  SZ(rmi_hip_shard);
  F(rmi_hip_shard, n_global); F(rmi_hip_shard, read_lo); F(rmi_hip_shard, read_hi); F(rmi_hip_shard, key_lo); F(rmi_hip_shard, key_hi);
  F(rmi_hip_shard, leaf_lo); F(rmi_hip_shard, leaf_hi); F(rmi_hip_shard, split_idx); F(rmi_hip_shard, split_target);
  SZ(rmi_hip_train_config);
  F(rmi_hip_train_config, root); F(rmi_hip_train_config, leaf_kind); F(rmi_hip_train_config, num_leaves);
  F(rmi_hip_train_config, root_table); F(rmi_hip_train_config, root_table_entries);
  SZ(rmi_hip_result);
  F(rmi_hip_result, num_rows); F(rmi_hip_result, num_leaves); F(rmi_hip_result, leaf_kind); F(rmi_hip_result, params_per_leaf);
  F(rmi_hip_result, row_bytes); F(rmi_hip_result, model_avg_error); F(rmi_hip_result, model_avg_l2_error);
  F(rmi_hip_result, model_avg_log2_error); F(rmi_hip_result, model_max_log2_error); F(rmi_hip_result, model_max_error);
  F(rmi_hip_result, model_max_error_idx); F(rmi_hip_result, split_idx); F(rmi_hip_result, split_target);
  F(rmi_hip_result, shard_leaf_lo); F(rmi_hip_result, shard_leaves); F(rmi_hip_result, sum_n_err); F(rmi_hip_result, sum_l2);
  F(rmi_hip_result, sum_log2); F(rmi_hip_result, device_ns); F(rmi_hip_result, kernel_ns); F(rmi_hip_result, long_leaves);
  F(rmi_hip_result, fit_mode_used); F(rmi_hip_result, merged_leaves); F(rmi_hip_result, exact_leaves); F(rmi_hip_result, guard_leaves);
  F(rmi_hip_result, generation);
  return 0;
}
