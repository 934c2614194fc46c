bash tools/gpu_call.sh r06zi rccl1
AB_VALUES="1 2 3" bash tools/gpu_call.sh r06zi ab:OMP355_LANES
AB_ARGS="--lanes 2" bash tools/gpu_call.sh r06zi ab:OMP355_LANE_SIDE
