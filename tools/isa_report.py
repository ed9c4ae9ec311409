#!/usr/bin/env python3
"""Static report over the gfx950 code objects inside libmoca_hip.so (no GPU needed: hipcc cross-compiles, llvm-objdump disassembles).

For every kernel: registers / spills / LDS from the code-object metadata, and for its MAIN LOOP (the innermost backward-branch loop
holding the most MFMAs): MFMAs, LDS-DMA loads (`buffer_load ... lds`), LDS fragment reads, barriers and -- the regression guard of
DESIGN 4.1 -- every `s_waitcnt vmcnt(0)` that sits between the first and the last MFMA of the loop body (a full drain of the DMA
stream inside the MFMA segment: what hipcc inserts when it cannot prove an LDS access disjoint from an LDS-DMA in flight, or when a
register that is a known load destination is rewritten).

    python tools/isa_report.py [libmoca_hip.so] [name filter ...]

`analyse(lib_path)` is what tests/test_isa_cpu.py calls."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(lib_path, workdir):
    """extract the gfx950 code objects of every translation unit bundled into the shared library"""
    tmp_lib = os.path.join(workdir, "lib.so")
    shutil.copy(lib_path, tmp_lib)                     # (llvm-objdump --offloading writes next to its input)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp_lib], check=True, capture_output=True, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f and f.endswith("gfx950"))


def metadata(co):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if line.lstrip().startswith("- ") and k in ("agpr_count", "args"):
            cur = {}
        if cur is None:
            continue
        if k == "name" and "kernel" in v and not v.endswith(".kd"):
            cur["name"] = v
            kernels[v] = cur
        elif k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                   "private_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = int(v)
    return kernels


_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def disassemble(co):
    """{mangled kernel name: [(addr, mnemonic, operands)]}"""
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = _INS.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def loops(ins):
    """backward branches -> (start index, end index) of loop bodies"""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    out = []
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>|<([^>+]+)>$", ops)
            tgt = None
            m2 = re.match(r"(\d+)", ops)
            if m and m.group(1):
                base = ins[0][0]
                tgt = base + int(m.group(1), 16)
            elif m2:                                   # raw simm16: target = next instruction + 4 * simm16
                off = int(m2.group(1))
                if off >= 0x8000:
                    off -= 0x10000
                tgt = a + 4 + 4 * off
            if tgt is not None and tgt <= a and tgt in addr_to_idx:
                out.append((addr_to_idx[tgt], i))
    return out


def loop_stats(ins, lo, hi):
    body = ins[lo:hi + 1]
    mf = [j for j, (_, mn, _) in enumerate(body) if mn.startswith("v_mfma")]
    st = dict(instructions=len(body), mfma=len(mf), lds_dma=0, ds_read=0, ds_write=0, barrier=0, vmcnt0_inside=0, vmcnt0=0, global_load=0, scratch=0)
    for j, (_, mn, ops) in enumerate(body):
        if (mn.startswith("buffer_load") and re.search(r"\blds\b", ops)) or "_lds_" in mn:
            st["lds_dma"] += 1
        elif mn.startswith(("buffer_load", "global_load")):
            st["global_load"] += 1
        elif mn.startswith("ds_read") or mn.startswith("ds_load"):
            st["ds_read"] += 1
        elif mn.startswith("ds_write") or mn.startswith("ds_store"):
            st["ds_write"] += 1
        elif mn == "s_barrier":
            st["barrier"] += 1
        elif mn.startswith("scratch_"):
            st["scratch"] += 1
        elif mn == "s_waitcnt" and re.search(r"vmcnt\(0\)", ops):
            st["vmcnt0"] += 1
            if mf and mf[0] < j < mf[-1]:
                st["vmcnt0_inside"] += 1
    return st


def main_loop(ins):
    """the steady-state loop: among the INNERMOST loops that hold MFMAs (a loop that encloses another MFMA loop -- a tile / group walk --
    is not one) -- those that also issue LDS-DMA, if any -- the one with the most MFMAs, shortest body first"""
    spans = [(lo, hi, loop_stats(ins, lo, hi)) for lo, hi in loops(ins)]
    spans = [sp for sp in spans if sp[2]["mfma"] > 0]
    inner = [sp for sp in spans if not any((o[0] >= sp[0] and o[1] <= sp[1] and (o[0], o[1]) != (sp[0], sp[1])) for o in spans)]
    cands = [sp[2] for sp in inner]
    if any(c["lds_dma"] for c in cands):
        cands = [c for c in cands if c["lds_dma"]]
    if not cands:
        return None
    top = max(c["mfma"] for c in cands)
    return min((c for c in cands if c["mfma"] == top), key=lambda c: c["instructions"])


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def analyse(lib_path=None):
    """{demangled kernel name: dict(metadata..., loop=dict(...) or None, mfma_total=...)} for every kernel of the library"""
    lib_path = lib_path or os.path.join(ROOT, "moca_video_amd", "libmoca_hip.so")
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(lib_path, wd):
            md, fn = metadata(co), disassemble(co)
            dm = demangle(list(md))
            for name, info in md.items():
                ins = fn.get(name)
                if ins is None:
                    continue
                d = dict(info)
                d["mfma_total"] = sum(1 for _, mn, _ in ins if mn.startswith("v_mfma"))
                d["instructions_total"] = len(ins)
                d["scratch"] = sum(1 for _, mn, _ in ins if mn.startswith("scratch_"))
                d["loop"] = main_loop(ins)
                short = re.sub(r"\(anonymous namespace\)::", "", dm[name])
                short = re.sub(r"^void ", "", short)
                short = re.sub(r"\(.*\)$", "", short)
                res[short] = d
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else None
    r = analyse(lib)
    print(f"{'kernel':46s} vgpr agpr sgpr spill | main loop: ins mfma dma dsrd barr vmcnt0(inside MFMA span) | mfma total")
    for k in sorted(r):
        if args and not any(a in k for a in args):
            continue
        d, lp = r[k], r[k]["loop"]
        ls = "-" if lp is None else f"{lp['instructions']:5d} {lp['mfma']:4d} {lp['lds_dma']:3d} {lp['ds_read']:4d} {lp['barrier']:4d} {lp['vmcnt0']:3d} ({lp['vmcnt0_inside']})"
        print(f"{k:46s} {d.get('vgpr_count', 0):4d} {d.get('agpr_count', 0):4d} {d.get('sgpr_count', 0):4d} "
              f"{d.get('vgpr_spill_count', 0) + d.get('sgpr_spill_count', 0):5d} | {ls} | {d['mfma_total']}")
