#!/bin/bash
# Collects the evidence kept under profiles/: kernel-trace statistics of the
# default bench run and the HBM traffic counters of its kernels (FETCH_SIZE and
# WRITE_SIZE in separate passes, as the counters do not fit one pass).
# usage (on the GPU box, from the repository root): tools/collect_profiles.sh <tag>
set -e
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python bench.py > $out/bench.json 2> $out/bench.err
# (--no-legs: the legs launch the same kernels on other frame sizes, which would mix
# into the per-kernel averages of the headline workload)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --no-cpu --no-legs > $out/bench_under_rocprof.json 2> $out/trace.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- python3 bench.py --no-cpu --no-reloc --no-legs --steps 3 --warmup 1 --streams 1 > /dev/null 2> $out/fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- python3 bench.py --no-cpu --no-reloc --no-legs --steps 3 --warmup 1 --streams 1 > /dev/null 2> $out/write.log
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $out/sq -o s --output-format csv -- python3 bench.py --no-cpu --no-reloc --no-legs --steps 3 --warmup 1 --streams 1 > /dev/null 2> $out/sq.log
python tools/pmc_summary.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $out/pmc_fetch_size_summary.csv
python tools/pmc_summary.py $(find $out/write -name "*counter_collection.csv" | head -1) $out/pmc_write_size_summary.csv
python tools/pmc_summary.py $(find $out/sq -name "*counter_collection.csv" | head -1) $out/pmc_sq_summary.csv
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
echo done
