# Round 6, VERDICT r05 item 2b: the guide's T5 (s_setprio) on the 8-wave 16-bit GEMM kernels, never A/B-ed before.
# prio16 = 0 none | 1 s_setprio 1/0 around each MFMA cluster incl. its interleaved fragment reads | 2 static s_setprio 1 for
# waves 4-7 | 3 s_setprio 1/0 around the bare MFMA cluster (fragment reads issued before it: the 8-phase template's form; q16s only)
set -x
O=gpurun_out
CONFIGS="base;prio16=1;prio16=2;prio16=3" T=131072,1048576 ROUNDS=3 FORWARD=1 python3 tools/gemm16_knob_ab.py > $O/r06_prio16_ab.log 2>&1
CONFIGS="base;prio16=1;prio16=2" MODE=f16x3 T=131072 ROUNDS=3 FORWARD=1 python3 tools/gemm16_knob_ab.py >> $O/r06_prio16_ab.log 2>&1
cat $O/r06_prio16_ab.log
