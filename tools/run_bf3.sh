set -x
cd /root/repo
python tools/probe_res_bf3.py 100000 256 64 64 > gpurun_out/bf3_probe.txt 2>&1
python tools/probe_res_bf3.py 10000 512 64 64 >> gpurun_out/bf3_probe.txt 2>&1
python tools/probe_res_bf3.py 40000 128 5 32 >> gpurun_out/bf3_probe.txt 2>&1
cat gpurun_out/bf3_probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "reservoir" 2>&1 | tail -5
