for ch in 1 4 8 16; do for cx in 1 2 4; do
python bench.py --workload config3 --config3-chunk $ch --contexts $cx --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c3_${ch}_${cx}.json
python - <<P
import json
d=json.load(open("gpurun_out/c3_${ch}_${cx}.json")); r=d["roofline"]; print("chunk $ch contexts $cx", d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("upscale_alone_ms"), r.get("avg_launch_ms_in_flow"), d["result_sample"])
P
done; done
