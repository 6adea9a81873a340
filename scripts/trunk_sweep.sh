for b in 4 8 12 16 24 32 48 64 96 128; do for w in 0 1; do
SE3TN_TRUNK_WINOGRAD=$w SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 60 --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%-4d trunk_wino=%d %9.1f pairs/s  %8.4f ms/step ' % ($b, $w, d['value'], d['ms_per_step']), ' '.join('%.3f'%v for v in list(d['layers_ms'].values())[2:6]))"
done; done
