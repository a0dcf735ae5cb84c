#!/bin/bash
# The library variants tools/r05_hunt.sh selects through FNR_LIB_PATH — build them HERE (CPU container: hipcc cross-compiles)
# before the gpurun call; the .so files travel with the snapshot.
set -e
cd "$(dirname "$0")/.."
bash tools/build_variant.sh seen -DFNR_SCATTER_DEBUG_SEEN
bash tools/build_variant.sh seen_nowait "-DFNR_SCATTER_DEBUG_SEEN -DFNR_SCATTER_FORCE_VECTOR_LOADS -DFNR_SCATTER_NO_COUNTER_WAIT"
bash tools/build_variant.sh seen_vec "-DFNR_SCATTER_DEBUG_SEEN -DFNR_SCATTER_FORCE_VECTOR_LOADS"
bash tools/build_variant.sh seen_atomic "-DFNR_SCATTER_DEBUG_SEEN -DFNR_SCATTER_ATOMIC_COUNTERS"
bash tools/build_variant.sh seen_rmw "-DFNR_SCATTER_DEBUG_SEEN -DFNR_SCATTER_RMW_COUNTERS"
bash tools/build_variant.sh atomic_counters -DFNR_SCATTER_ATOMIC_COUNTERS
bash tools/build_variant.sh rmw_counters -DFNR_SCATTER_RMW_COUNTERS
bash tools/build_variant.sh zero_early -DFNR_ACC_ZERO_EARLY    # A/B only (tools/r05_ab_counters.sh): accumulator zeroed while the counter loads fly, one barrier fewer
