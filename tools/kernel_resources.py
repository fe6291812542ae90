"""Registers, spills and LDS of every kernel in modelcompose_amd/libmc_hip.so, from the code objects' own metadata (no recompilation):
the .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled for gfx950 and its AMDGPU metadata note read.

    python tools/kernel_resources.py [substring ...]        # JSON lines: name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_resources(so_path=None):
    so_path = so_path or os.path.join(ROOT, "modelcompose_amd", "libmc_hip.so")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.devnull], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, st in enumerate(starts):
            part = os.path.join(tmp, f"b{i}.bin")
            open(part, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"co{i}.o")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                line = line.strip()
                m = re.match(r"- \.(\w+):\s*(.*)$", line) or re.match(r"\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" or (k == "args" and cur is None):
                    pass
                if line.startswith("- .") and k in ("agpr_count", "args"):
                    cur = {}
                    out.append(cur)
                if cur is None:
                    continue
                if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                         "private_segment_fixed_size", "max_flat_workgroup_size"):
                    cur[k] = v if k == "name" else int(v)
    return [k for k in out if "name" in k and "vgpr_count" in k]


if __name__ == "__main__":
    pats = sys.argv[1:]
    for k in sorted(kernel_resources(), key=lambda k: k["name"]):
        if not pats or any(p in k["name"] for p in pats):
            print(json.dumps(k))
