for n in 128 256; do for l in 1 2 4; do
  timeout -k 10 200 python bench.py --mode p3m --n-side $n --no-cpu-baseline --steps 5 --p3m-lanes $l 2>&1 | tail -1 > gpurun_out/lanes_${n}_${l}.json || exit 1
done; done
