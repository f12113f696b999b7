#!/bin/bash
# The round's side measurements (everything DESIGN.md quotes besides the default bench line and the rocprof profile):
#   gpurun -- 'bash tools/measure_round.sh r02d'   ->  gpurun_out/meas_<tag>/*.json   (copy into profiles/<tag>_measurements.json with
#   tools/measure_round.sh's last step, which merges the one-line results)
tag=${1:-r02}
ulimit -c 0
out=gpurun_out/meas_$tag
mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs"
# one slab of the strong-scaling split of C3 on a device of its own (what each GPU of an N-GPU run computes, without exchange)
for nx in 64 32 16; do $B --nx $nx --steps 100 > $out/slab_nx$nx.json 2> $out/slab_nx$nx.err; done
# the same slabs as ONE RANK of the one-object-per-rank path whose neighbours are the rank itself: the whole exchange (pack, RCCL send and
# receive on the engine's stream, overlapped re-sort, unpack) on top of the slab's own step
for nx in 128 64 32 16; do $B --nx $nx --self-ring --steps 192 --warmup 8 > $out/selfring_nx$nx.json 2> $out/selfring_nx$nx.err; done
# the native multi_HIP object with all slabs on this one device: concurrent (exchange overlapped) and one slab at a time
for N in 2 4 8; do
  $B --gpus $N --oversubscribe --steps 40 > $out/multi_${N}_concurrent.json 2> $out/multi_${N}_concurrent.err
  $B --dbg MULTI_SERIALIZE --gpus $N --oversubscribe --steps 40 > $out/multi_${N}_serialized.json 2> $out/multi_${N}_serialized.err
done
# C4 at full size (256 x 256 x 128 x 64 = 5.4e8 SDs, 8 slabs of 32 x-planes), all slabs on this device: concurrently, and one slab at a time
C4="--nx 256 --ny 256 --nz 128 --gpus 8 --oversubscribe --steps 10 --warmup 2"
$B $C4 > $out/c4_8slabs_concurrent.json 2> $out/c4_8slabs_concurrent.err
$B --dbg MULTI_SERIALIZE $C4 > $out/c4_8slabs_serialized.json 2> $out/c4_8slabs_serialized.err
# one C4 slab alone on the device (no neighbours)
$B --nx 32 --ny 256 --nz 128 --steps 40 > $out/c4_slab_alone.json 2> $out/c4_slab_alone.err
# C5: 128^3 x 512 SD/cell (66 steps: four storage re-orderings at the crowded cells' period of 16)
$B --sd-conc 512 --steps 66 --warmup 2 > $out/c5.json 2> $out/c5.err
# coalescence that collides (the Golovin test's spectrum), 64 and 512 per cell
$B --workload coal-stress --steps 100 > $out/coal_stress.json 2> $out/coal_stress.err
$B --workload coal-stress --sd-conc 512 --steps 34 --warmup 2 > $out/coal_stress_512.json 2> $out/coal_stress_512.err
# the reference's TOMS748 iterates in fast arithmetic as the headline configuration (the bench line carries it as a 20-step leg)
$B --cond-solver toms748 --steps 100 > $out/cond_solver_toms748.json 2> $out/cond_solver_toms748.err
# opts_init.stream_ordered on the 16-plane slab with its exchange, and on the whole box
$B --nx 16 --self-ring --steps 192 --warmup 8 --stream-ordered 1 > $out/selfring_nx16_stream_ordered.json 2> $out/selfring_nx16_stream_ordered.err
$B --steps 200 --stream-ordered 1 > $out/default_stream_ordered.json 2> $out/default_stream_ordered.err
# what the side stream and the bucket ranking are worth (the whole box)
$B --steps 200 --dbg NO_RANK_OVERLAP > $out/default_no_rank_overlap.json 2> $out/default_no_rank_overlap.err
$B --steps 200 --dbg NO_RANK_OVERLAP,RANK_BY_COUNTING,FINISH_STAGED > $out/default_round4_midway.json 2> $out/default_round4_midway.err
python3 - "$out" "$tag" <<'PY'
import glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        res[os.path.basename(f)[:-5]] = {"error": str(e)}
        continue
    res[os.path.basename(f)[:-5]] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "n_gpus": d["n_gpus"], "steps": d["steps"],
                                      "workload": d["config"]["workload"], "decomposition": d["config"]["decomposition"],
                                      "stage_ms_per_step": d.get("stage_ms_per_step")}
json.dump(res, open(os.path.join(out, "measurements.json"), "w"), indent=1)
for k, v in res.items():
    print(k, v.get("ms_per_step"), v.get("value"))
PY
