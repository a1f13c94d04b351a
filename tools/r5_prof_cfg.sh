# kernel stats of one config:  bash tools/r5_prof_cfg.sh cfg2 [ENV=VALUE ...]   (through gpurun; writes gpurun_out/r5p_<cfg>_kernel_stats.csv)
cfg=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r5p_$cfg -o p --output-format csv -- python3 /root/repo/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/r5p_$cfg.log 2>&1
cp $(find /root/repo/gpurun_out/r5p_$cfg -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r5p_${cfg}_kernel_stats.csv
rm -rf /root/repo/gpurun_out/r5p_$cfg
tail -1 /root/repo/gpurun_out/r5p_$cfg.log | cut -c1-300
