for i in 1 2 3 4 5 6 7 8; do python tools/probe/cold_start_probe.py 1024 200 | head -1; done
echo "--- with torch initialised first"
for i in 1 2 3 4 5 6; do PROBE_TORCH=1 python tools/probe/cold_start_probe.py 1024 200 | head -1; done
