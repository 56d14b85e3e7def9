#!/bin/bash
# round 5, GPU call A: doc / boundary tests + the gemm256 experiments (one twin library each) against the product kernel
out=gpurun_out/${1:-r05a}; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests/test_integration_doc.py tests/test_c_abi_demo.py tests/test_gpu_ops.py -m gpu -x -q -k "integration or c_abi or vit_create or set_params" > $out/pytest_doc.log 2>&1
tail -3 $out/pytest_doc.log
fi
for tag in ${TAGS:-base pipe sc1 nt pipesc1}; do
    extra=""; case $tag in base|nt|${TIMING_ONLY_TAG:-none}) extra="--timing-only";; esac
    ATLASPATCH_HIP_LIB=atlaspatch_amd/libatlaspatch_hip_twin_$tag.so timeout 900 python tools/gemm_twin_ab.py $extra > $out/twin_$tag.log 2>&1
    echo "== $tag rc=$?"; grep -E "bit-equality|repeatability|FAIL" $out/twin_$tag.log | head -8
    python - <<PY
import json
r=json.loads(open("$out/twin_$tag.log").read().strip().splitlines()[-1])
by={}
for x in r: by.setdefault(x["gemm"],{})[(x["impl"],x["ablate"])]=x["ms_median"]
for g,d in by.items(): print("   %-16s"%g, "  ".join("%s:%.4f"%(k,v) for k,v in d.items()), " twin/product %.4f"%(d[(257,0)]/d[(256,0)]))
PY
done
for tag in ${DTAGS:-diag diagpipe}; do
    ATLASPATCH_HIP_LIB=atlaspatch_amd/libatlaspatch_hip_twin_$tag.so timeout 900 python tools/gemm_twin_ab.py --timing-only --ablate ${ABL:-8,4,12} > $out/twin_$tag.log 2>&1
    echo "== $tag rc=$?"
    python - <<PY
import json
r=json.loads(open("$out/twin_$tag.log").read().strip().splitlines()[-1])
by={}
for x in r: by.setdefault(x["gemm"],{})[(x["impl"],x["ablate"])]=x["ms_median"]
for g,d in by.items(): print("   %-16s"%g, "  ".join("%s:%.4f"%(k,v) for k,v in d.items()))
PY
done
