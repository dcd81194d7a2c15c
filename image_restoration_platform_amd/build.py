"""In-tree build of libire.so with hipcc for gfx950 (no cmake/ninja; seconds per file).

`python -m image_restoration_platform_amd.build` or `__graft_entry__.build()`.
The .so stays in-tree (git-ignored) so it travels to the GPU box with the snapshot.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libire.so")

# (source, extra flags).  classifier.hip carries the IEEE-exact double finalize: no FMA contraction.
SOURCES = [
    ("classifier.hip", ["-ffp-contract=off"] + (["-DCLS_ABL=" + os.environ["CLS_ABL"]] if os.environ.get("CLS_ABL") else [])),
    ("conv_mfma.hip", []),
    ("conv_rb.hip", (["-DIRE_RB_ABLATE"] if os.environ.get("IRE_RB_ABLATE") else []) +
     (["-DIRE_RB_NGP64=" + os.environ["IRE_RB_NGP64"]] if os.environ.get("IRE_RB_NGP64") else [])),
    ("conv_w4.hip", (["-DIRE_W4_STAMPS"] if os.environ.get("IRE_RB_ABLATE") else []) +
     (["-DIRE_W4_TICKS"] if os.environ.get("IRE_RB_ABLATE") == "2" else []) +
     (["-DW4_TEPI=" + os.environ["IRE_W4_TEPI"]] if os.environ.get("IRE_W4_TEPI") else []) +
     (["-DIRE_W4_TL"] if os.environ.get("IRE_W4_TL") else [])),
    ("conv_up.hip", (["-DIRE_UP_ABL=" + os.environ["IRE_UP_ABL"]] if os.environ.get("IRE_UP_ABL") else []) +
     (["-DIRE_UP_D=" + os.environ["IRE_UP_D"]] if os.environ.get("IRE_UP_D") else []) +
     (["-DIRE_UP_TEPI=" + os.environ["IRE_UP_TEPI"]] if os.environ.get("IRE_UP_TEPI") else [])),
    ("conv_upq.hip", os.environ.get("UQ_DEFS", "").split()),
    ("conv_dnq.hip", os.environ.get("DQ_DEFS", "").split()),
    ("conv_down.hip", (["-DDN_LD_NT=" + os.environ["DN_LD_NT"]] if os.environ.get("DN_LD_NT") else [])),
    ("conv_stem.hip", []),
    ("conv_f8.hip", []),
    ("conv_pc.hip", (["-DC3_ABL=" + os.environ["C3_ABL"]] if os.environ.get("C3_ABL") else []) +
     (["-DC3_PRIO=" + os.environ["C3_PRIO"]] if os.environ.get("C3_PRIO") else []) +
     (["-DC3_RES_PRE=" + os.environ["C3_RES_PRE"]] if os.environ.get("C3_RES_PRE") else []) +
     (["-DC3_TEPI=" + os.environ["C3_TEPI"]] if os.environ.get("C3_TEPI") else []) +
     (["-DC3_TEPI64=" + os.environ["C3_TEPI64"]] if os.environ.get("C3_TEPI64") else []) +
     (["-DIRE_LD_ONCE32=" + os.environ["IRE_LD_ONCE32"]] if os.environ.get("IRE_LD_ONCE32") else []) +
     (["-DC3_PROD8=" + os.environ["C3_PROD8"]] if os.environ.get("C3_PROD8") else []) +
     (["-DIRE_PC_TICKS"] if os.environ.get("IRE_RB_ABLATE") == "2" else []) + (["-DIRE_W4_TL"] if os.environ.get("IRE_W4_TL") else [])),
    ("conv_pk.hip", (["-DPK_ABL=" + os.environ["PK_ABL"]] if os.environ.get("PK_ABL") else []) +
     (["-DPK_TICKS"] if os.environ.get("PK_TICKS") else []) + (["-DIRE_W4_TL"] if os.environ.get("IRE_W4_TL") else []) + os.environ.get("PK_DEFS", "").split()),
    ("gn.hip", []),
    ("fusion.hip", (["-DFUSE_FL=" + os.environ["FUSE_FL"]] if os.environ.get("FUSE_FL") else [])),
    ("preprocess.hip", []),
    ("encode.hip", []),
    ("engine.cpp", []),
    ("strips.cpp", []),
    ("api.cpp", []),
]
# -fno-slp-vectorize: hipcc otherwise packs adjacent f32 adds / muls / FMAs into v_pk_*_f32, which beside MFMAs cost several
# times two plain instructions (MI355X_MICROARCH.md per-instruction constants; measured: profiles/r02_experiments.md).
# IRE_SLP=1 builds with the vectorizer on (A/B).
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-D__HIP_PLATFORM_AMD__"] + ([] if os.environ.get("IRE_SLP") == "1" else ["-fno-slp-vectorize"]) + \
         (["-DIRE_ST_LINE=" + os.environ["IRE_ST_LINE"]] if os.environ.get("IRE_ST_LINE") else []) + \
         (["-DIRE_ST_PART=" + os.environ["IRE_ST_PART"]] if os.environ.get("IRE_ST_PART") else []) + \
         (["-DIRE_LD_ONCE=" + os.environ["IRE_LD_ONCE"]] if os.environ.get("IRE_LD_ONCE") else []) + \
         (["-DIRE_LD_IN=" + os.environ["IRE_LD_IN"]] if os.environ.get("IRE_LD_IN") else []) + \
         os.environ.get("IRE_XFLAGS", "").split()          # extra compiler flags for build-time A/B (tools/s2_xflags.sh)

def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps_mtime():
    m = 0.0
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hpp", ".inc", ".h")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    m = max(m, os.path.getmtime(os.path.join(HERE, "..", "include", "ire.h")))
    return m


def _compile(src, extra, force, hdr_m):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".", "_") + ".o")
    lang = ["-x", "hip"] if src.endswith((".hip", ".cpp")) else []
    cmd = [_hipcc()] + COMMON + extra + lang + ["-c", path, "-o", obj]
    stamp = obj + ".cmd"      # the command line the object was built with: a changed flag (build-time A/B macros) rebuilds it
    cmd_id = " ".join(cmd).replace(HERE, "<pkg>")      # (location-independent: the tree is built here and used on the GPU box)
    same_cmd = os.path.exists(stamp) and open(stamp).read() == cmd_id
    if (not force and same_cmd and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(path)
            and os.path.getmtime(obj) > hdr_m):
        return obj
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(cmd_id)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [(s, f) for s, f in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if os.path.exists(os.path.join(CSRC, "fusion_stub.cpp")) and not os.path.exists(os.path.join(CSRC, "fusion.hip")):
        srcs.append(("fusion_stub.cpp", []))
    hdr_m = _deps_mtime()
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda sf: _compile(sf[0], sf[1], force, hdr_m), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    build_node_addon()
    try:
        # the PyTorch-ROCm host is one of three hosts of the same C ABI: a box without torch (the numpy / ctypes host, the Node
        # shim) still gets libire.so; torch_host.load_extension fails loudly when the extension is then asked for
        build_torch_extension()
    except ImportError as e:
        print("build: torch extension skipped (%s)" % e, file=sys.stderr)
    except RuntimeError as e:
        print("build: torch extension NOT built: %s" % str(e)[-1500:], file=sys.stderr)
    if verbose:
        print("built", LIB)
    return LIB


def build_node_addon():
    """N-API shim for the server-node worker (plain g++; N-API symbols resolve from the node binary)."""
    src = os.path.join(HERE, "node", "ire_napi.cc")
    out = os.path.join(HERE, "node", "ire_napi.node")
    inc = "/usr/include/node"
    if not os.path.exists(os.path.join(inc, "node_api.h")):
        return None   # no node headers on this host: the Python/ctypes host is unaffected
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), _deps_mtime()):
        return out
    r = subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I" + inc, "-DNODE_GYP_MODULE_NAME=ire_napi", src, "-o", out, "-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("node addon build failed:\n" + r.stderr[-2000:])
    return out


def build_torch_extension():
    """The PyTorch-ROCm extension host (csrc/torch_ext.cpp -> lib/_ire_torch.so), built in-tree with the include / library
    paths torch.utils.cpp_extension reports; plain C++ (no device code): g++.  It dlopens libire.so at run time."""
    src = os.path.join(CSRC, "torch_ext.cpp")
    out = os.path.join(HERE, "lib", "_ire_torch.so")
    import sysconfig
    from torch.utils import cpp_extension as ce
    incs = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    libdir = ce.library_paths()[0]
    cmd = (["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-DTORCH_EXTENSION_NAME=_ire_torch",
            "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(__import__("torch")._C._GLIBCXX_USE_CXX11_ABI))] +
           ["-isystem" + i for i in incs] + [src, "-o", out, "-L" + libdir, "-Wl,-rpath," + libdir,
            "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-ldl"])
    stamp = os.path.join(OBJ, "_ire_torch.cmd")       # rebuilt when the source, the header or the command line (flags, torch paths) changed
    cmd_id = " ".join(cmd).replace(HERE, "<pkg>")
    if (os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == cmd_id and
            os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "include", "ire.h")))):
        return out
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch extension build failed:\n" + r.stderr[-3000:])
    os.makedirs(OBJ, exist_ok=True)
    with open(stamp, "w") as f:
        f.write(cmd_id)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
