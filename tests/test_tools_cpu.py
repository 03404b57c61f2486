"""CPU: the analysis tools that produce tables quoted in DESIGN.md run on the committed profile data."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rnd", ["r02", "r03", "r04", "r05"])
def test_critical_path_on_the_committed_traces(rnd):
    for tag in ("default", "onestream", "zinc", "chembl"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "critical_path.py"),
                              os.path.join(ROOT, "profiles", rnd, f"trace_{tag}.csv"), "0"],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        rows = {l[:28].strip(): l[28:].split() for l in out.stdout.splitlines()[1:]}
        assert "bwd: message passes" in rows and "step (from compact_fill)" in rows
        span, main, side = (float(x) for x in rows["step (from compact_fill)"])
        assert 1500 < span < 8000 and main <= span + 1
        if tag == "onestream":
            assert side == 0.0                      # weight gradients on the main queue
        else:
            assert side > 0.2 * span                # a second queue overlaps the backward
        if rnd in ("r04", "r05") and tag == "default":   # since round 4: nothing left behind the last dZ chain (155 us in r03)
            assert float(rows["tail (side queue only)"][0]) < 20.0
            committed = open(os.path.join(ROOT, "profiles", rnd, "critical_path_default.txt")).read()
            full = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "critical_path.py"),
                                   os.path.join(ROOT, "profiles", rnd, "trace_default.csv")], capture_output=True, text=True)
            assert full.stdout.strip() == committed.strip()           # the committed table is this tool's output


def test_gemm_class_report_on_the_committed_trace_and_launch_log():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_class_report.py"),
                          os.path.join(ROOT, "profiles", "r02", "trace_onestream.csv"),
                          os.path.join(ROOT, "profiles", "r02", "gemm_launch_log.txt")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    hidden = [float(l.split()[-1]) for l in lines if l.startswith("forward  4 problem(s)") and "K <= 500" in l
              and "N <= 500" in l]
    assert len(hidden) == 3 and all(0.5 < f < 0.75 for f in hidden)       # 0.60-0.62 of peak
    wgrad = float([l for l in lines if l.startswith("wgrad")][0].split()[-1])
    assert 0.4 < wgrad < 0.65


def test_gemm_class_report_round_3_with_the_bf16x3_launches():
    """The r03 launch log carries the bf16x3 launches as classes b0 / b1 (gi_gemm_bf3.hip); the report matches them
    with the gi_gemm_bf3_kernel<...> rows of the trace and reproduces the figures of DESIGN.md section 5."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_class_report.py"),
                          os.path.join(ROOT, "profiles", "r03", "trace_onestream.csv"),
                          os.path.join(ROOT, "profiles", "r03", "gemm_launch_log.txt")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    fwd = [float(l.split()[-2]) for l in lines if l.startswith("forward bf16x3")]     # frac of the fp32 MFMA peak
    dgr = [float(l.split()[-2]) for l in lines if l.startswith("dgrad bf16x3")]
    own = [float(l.split()[-1]) for l in lines if l.startswith(("forward bf16x3", "dgrad bf16x3"))]
    assert len(fwd) == 3 and len(dgr) == 3
    assert all(0.75 < f < 1.0 for f in fwd) and all(0.65 < f < 0.95 for f in dgr)     # 0.87-0.88 / 0.78-0.80
    assert all(0.25 < f < 0.40 for f in own)                                          # against 397: 0.31-0.35
    committed = open(os.path.join(ROOT, "profiles", "r03", "gemm_class_report.txt")).read()
    assert out.stdout.strip() == committed.strip()                                     # the committed report is this output


def test_gemm_class_report_round_4_with_the_fp16x2_launches():
    """The r04 launch log carries the fp16x2 launches as classes x0 / x1 (gi_gemm_bf3.hip) and y2 (gi_gemm_b3p.hip, weight
    gradients); the report prices them against the fp32 MFMA peak AND against their own pipe (2382 / 3 TFLOP/s)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_class_report.py"),
                          os.path.join(ROOT, "profiles", "r04", "trace_onestream.csv"),
                          os.path.join(ROOT, "profiles", "r04", "gemm_launch_log.txt")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    fwd = [l.split() for l in lines if l.startswith("forward fp16x2")]
    dgr = [l.split() for l in lines if l.startswith("dgrad fp16x2")]
    assert len(fwd) == 3 and len(dgr) == 3
    for row in fwd + dgr:
        frac, own = float(row[-2]), float(row[-1])
        assert 0.85 < frac < 1.3 and 0.15 < own < 0.30 and abs(own - frac * 157.3 / 794.0) < 0.011
    w16 = [l.split() for l in lines if "16-bit pipe (fp16x2)" in l]
    assert len(w16) == 1 and float(w16[0][-2]) > 1.2 and 0.2 < float(w16[0][-1]) < 0.4
    committed = open(os.path.join(ROOT, "profiles", "r04", "gemm_class_report.txt")).read()
    assert out.stdout.strip() == committed.strip()


def test_chain_scaling_report_reads_a_kernel_trace(tmp_path):
    """tools/chain_scaling.py report: launches grouped by kernel and number of workgroups, median of the last four."""
    cols = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id",
            "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
            "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]
    lines = [",".join('"%s"' % c for c in cols)]
    t = 1000
    for name, us in (("void (anonymous namespace)::gi_chain_x2r_kernel<false, true>((anonymous namespace)::ChainArgs)", 55),
                     ("void (anonymous namespace)::gi_chain_kernel<false, 1, 2>((anonymous namespace)::ChainArgs)", 103),
                     ("(anonymous namespace)::gi_chain_pack_x2_kernel((anonymous namespace)::PackArgs)", 5)):
        for rep in range(6):
            dur = us * 1000 + (500000 if rep < 2 else rep)          # two slow warm-up launches, ignored by the median
            row = ["KERNEL_DISPATCH", 1, 1, 1, 1, 1, 1, name, 1, t, t + dur, 0, 0, 0, 0, 0, 512, 1, 1, 264 * 512, 1, 1]
            lines.append(",".join('"%s"' % v for v in row))
            t += dur + 1000
    path = tmp_path / "trace.csv"
    path.write_text("\n".join(lines) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "chain_scaling.py"), "report", str(path)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    row = out.stdout.splitlines()[-1].split()
    assert row[0] == "264" and abs(float(row[1]) - 103.0) < 0.1 and row[2] == "-" and abs(float(row[3]) - 55.0) < 0.1, out.stdout
