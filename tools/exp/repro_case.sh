cd /root/repo
for P in f16x3 f32; do echo == $P; FUZZ_ONLY=5 FUZZ_PREC=$P timeout 300 python tools/fuzz_fused.py 16 7 2>&1 | grep -E "^ok|FAIL|failures" | cut -c1-420; done
echo == f16x3 with the fp32 backward kernel
STEGO_DEBUG_BWD=512 FUZZ_ONLY=5 FUZZ_PREC=f16x3 timeout 300 python tools/fuzz_fused.py 16 7 2>&1 | grep -E "^ok|FAIL|failures" | cut -c1-420
echo == f16x3 three-launch forward
STEGO_FWD_VARIANT=1 FUZZ_ONLY=5 FUZZ_PREC=f16x3 timeout 300 python tools/fuzz_fused.py 16 7 2>&1 | grep -E "^ok|FAIL|failures" | cut -c1-420
