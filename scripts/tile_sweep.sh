for b in 6 8 12 16 24 32 48; do for t in 4 6; do
SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 30 --streams 1 --winograd 6 --winograd-tile $t 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%-3d F(%dx%d) %9.1f pairs/s %8.4f ms/step' % ($b, $t, $t, d['value'], d['ms_per_step']))"
done; done
