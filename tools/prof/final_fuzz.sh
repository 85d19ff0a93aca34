# the randomised parity sweeps on the final tree of a round (outputs: gpurun_out/fuzz_final/)
mkdir -p gpurun_out/fuzz_final
cd tests/perf
timeout 1500 python fuzz_parity.py 300 601 > ../../gpurun_out/fuzz_final/fuzz_parity_300_seed601.txt 2>&1
FUZZ_LARGE=1 timeout 900 python fuzz_parity.py 30 602 > ../../gpurun_out/fuzz_final/fuzz_parity_large_30_seed602.txt 2>&1
timeout 900 python fuzz_marginalize.py 500 603 > ../../gpurun_out/fuzz_final/fuzz_marginalize_500_seed603.txt 2>&1
for s in 61 62 63 64; do timeout 900 python fuzz_composite.py 60 $s > ../../gpurun_out/fuzz_final/fuzz_composite_60_seed$s.txt 2>&1; done
cd ../..
tail -n 1 gpurun_out/fuzz_final/*.txt
