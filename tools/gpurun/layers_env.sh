# per-layer tables only (no bench): BATCH=8 bash tools/gpurun/layers_env.sh "ENV_A" "ENV_B" ... ; prints rows matching $GREP
BATCH=${BATCH:-8} bash tools/gpurun/ab_layers_env.sh "$@" 2>&1 | grep -E "${GREP:-.}" | cut -c1-150
