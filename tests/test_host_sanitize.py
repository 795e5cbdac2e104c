"""CPU tier: the product's host-side table builders under AddressSanitizer + UBSan."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_tables_under_sanitizers(tmp_path):
    csrc = os.path.join(REPO, "foveated-360-video_amd", "csrc")
    exe = str(tmp_path / "host_tables_sanitize")
    # host_tables.cpp also holds the C-ABI exports (they need HIP headers): compile only the
    # table builders by cutting the file at the marker
    text = open(os.path.join(csrc, "host_tables.cpp")).read()
    cut = text.index("// ---- C ABI: host-only exports")
    src = tmp_path / "host_tables_only.cpp"
    src.write_text(text[:cut])
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-ffp-contract=off", "-I", csrc, str(src),
                    os.path.join(REPO, "tests", "native", "host_tables_sanitize.cc"), "-o", exe],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("ok ")
