"""The host key index (open addressing, backward-shift deletion) against std::unordered_map, on the CPU."""

import os
import subprocess


def test_keymap_randomised_against_unordered_map(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "keymap_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(here, "keymap_check.cpp")], check=True)
    # the timeout also guards against hash clustering: the whole check takes a few seconds when probing is O(1)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "keymap ok" in out.stdout
