set -x
cd /root/repo
timeout 300 tools/gemm_lab.bin 10 ':pp:,:pp16:' > gpurun_out/pp16_lab.txt 2>&1
tail -80 gpurun_out/pp16_lab.txt
