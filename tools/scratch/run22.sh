for t in 64 128; do for w in c3 c5; do
echo "tile $t workload $w"
VGAN_GRAM_TILE=$t python bench.py --workload $w --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print(round(j['value'],1),'steps/s', r['kernel'], round(r['avg_launch_ms']*1e3,1),'us', round(r['achieved'],1),'TF')"
done; done
