B=262144
K1_MICRO_PER_BUFFER=1 python tools/k1_micro.py --batch $B --reps 100 --perm
NFLOWS_AMD_LIB=build_variants/wt_ntio.so python tools/k1_micro.py --batch $B --reps 100 --perm
python tools/k1_micro.py --batch $B --reps 100 --perm --inverse
python tools/k1_micro.py --batch 65536 --reps 100 --perm
NFA_K1_WAVETILE=0 python tools/k1_micro.py --batch 65536 --reps 100 --perm
python tools/k1_micro.py --batch $B --reps 100 --bins 10
NFA_K1_WAVETILE=0 python tools/k1_micro.py --batch $B --reps 100 --bins 10
