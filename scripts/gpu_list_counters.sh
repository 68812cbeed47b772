mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters.txt 2>&1
grep -c . gpurun_out/counters.txt
grep -oE "Name:\s*\S+|^\s*[A-Z][A-Za-z0-9_]+\s" gpurun_out/counters.txt | head -5
