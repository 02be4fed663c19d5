for rep in 1 2; do
for k in dma pipe; do
  for cfg in cfg2 cfg5_shard; do
    v=$(FA_AMD_KERNEL=$k timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg 2>/dev/null | grep -o '"value": [0-9.]*')
    echo "$rep $k $cfg $v"
  done
done
done
