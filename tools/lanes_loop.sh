mkdir -p gpurun_out/r06zd
for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests -m gpu -q -k "pipelined_lanes" 2>&1 | tail -1; done > gpurun_out/r06zd/lanes_loop.txt
timeout 900 python -m pytest tests -m gpu -q -k "graph or contexts or lanes or eos or decoder_fused or longest" 2>&1 | tail -2 >> gpurun_out/r06zd/lanes_loop.txt
cat gpurun_out/r06zd/lanes_loop.txt
