# VERDICT r05 item 6 priced on the K-loop model of the split-bf16 GEMM (tools/micro/split_loop_model2.hip)
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/micro/split_loop_model2.hip -o /tmp/slm2 2>/dev/null && /tmp/slm2 > gpurun_out/r06_presplit_loop_model.txt 2>&1
/tmp/slm2 >> gpurun_out/r06_presplit_loop_model.txt 2>&1
cat gpurun_out/r06_presplit_loop_model.txt | tail -24
