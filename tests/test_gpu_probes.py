"""Device-side probes, collected by the GPU suite (round 4).

tests/_devprobes/*.hip (+ tools/pkf32_probe.hip) are small HIP programs built by __graft_entry__.build() into
tests/_devprobes/bin/ (the binaries travel with the snapshot like the library).  They are test infrastructure: nothing
in the product loads them.

  * convert_pair_probe: nfa::k8h::convert_pair -- the inline-asm piece conversion of the f16 whole-layer kernels, called
    from the product's own header -- against the plain convert / subtract / convert sequence, 4 M pairs per
    configuration, magnitudes on both sides of the f16 range: bit for bit (host-compiled arithmetic tests cannot cover
    a function that IS its instructions).
  * mfma_dst_on_src_probe / pkf32_probe: reproducers of the two compiler / hardware hazards the f16 kernels work
    around.  What is REQUIRED: the forms the product uses are clean.  What is RECORDED (and
    compared with the behaviour written down here, so that a driver / compiler / chip revision that changes it is
    seen): the count of wrong results of the forms the product avoids.
"""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_devprobes", "bin")


def _run(name, *args, timeout=300):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.fail("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % exe)
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    return r.returncode, r.stdout + r.stderr


def _report(entry):
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "probe_report.jsonl"), "a") as f:
            f.write(json.dumps(entry) + "\n")
    except OSError:
        pass
    print("\n[probe] " + json.dumps(entry))


def test_piece_conversion_of_the_f16_kernels_matches_the_plain_sequence():
    rc, text = _run("convert_pair_probe")
    lines = [ln for ln in text.splitlines() if ln.startswith("convert_pair")]
    _report({"probe": "convert_pair", "lines": lines})
    assert rc == 0 and "convert_pair_probe: OK" in text, text[-2000:]
    assert len(lines) == 13        # 3 scales x (relu, guard) + the verdict


def test_mfma_result_on_its_own_operand_reproducer():
    """1 000 chains of eight v_mfma_f32_16x16x32_f16 per wave, two waves per SIMD on every CU: the SAFE form (result on
    registers of its own: what the kernels' keep-alive guarantees) must be bit-identical to the padded reference chain;
    the OVERLAP form (result on the A fragment's registers: what hipcc emitted for K8s before the guard) is run and its
    count recorded."""
    rc, text = _run("mfma_dst_on_src_probe", 1000)
    got = dict(re.findall(r"mfma_dst_on_src (\w+): (\d+) wrong", text))
    _report({"probe": "mfma_dst_on_src", "wrong_lane_results": got, "iterations": 1000})
    assert set(got) == {"overlap", "safe"}, text[-2000:]
    assert int(got["safe"]) == 0, text[-2000:]


def test_packed_fp32_beside_an_mfma_wave_reproducer():
    """v_pk_add_f32 / v_pk_mul_f32 consumed 0 / 1 / 4 instructions later, alone and beside a wave that keeps the matrix
    pipe busy, 20 000 iterations on every CU.  In K8h packed fp32 conversions gave wrong lanes 16-31 (the MFMA files are
    built with -fno-slp-vectorize since); this isolated form has so far stayed clean on every box -- recorded, and
    required to stay so: a non-zero count is the first reproducer of the hazard and must be looked at."""
    rc, text = _run("pkf32_probe")
    counts = [int(c) for c in re.findall(r"wrong results\s+(\d+)", text)]
    _report({"probe": "pkf32", "wrong_results_per_configuration": counts})
    assert len(counts) == 8, text[-2000:]
    assert sum(counts) == 0, text[-2000:]
