#!/bin/bash
export TMPDIR=/tmp
python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 > gpurun_out/r3_p_irv2_bench.json 2>/dev/null; cat gpurun_out/r3_p_irv2_bench.json
python bench.py --backbone InceptionResNetV2 --mode predict --batch 16 --steps 50 --warmup 5 --no-kernel-timers 2>/dev/null
python tools/var/irv2_host.py 2>&1 | grep -E "steps:|idle GPU"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_f_irv2 -- python3 bench.py --backbone InceptionResNetV2 --batch 16 --steps 5 --warmup 2 --no-kernel-timers > gpurun_out/r3_p_irv2_under_rocprof.json 2> gpurun_out/r3_p_irv2.err; echo "irv2 prof rc=$?"
