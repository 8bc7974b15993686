"""Compile libgda_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the built library
sits next to this file so that it travels with the source tree."""
import concurrent.futures as cf
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgda_hip.so")
SOURCES = ["gda_graph.hip", "gda_spmm.hip", "gda_kstep.hip", "gda_mmd.hip", "gda_disc.hip", "gda_disc_mlp.hip", "gda_critic.hip", "gda_mixup.hip", "gda_misc.hip", "gda_gat.hip", "gda_act.hip", "gda_attention.hip",
           "gda_laplacian.hip", "gda_optim.hip", "gda_gemm.hip", "gda_ce.hip", "gda_ppmi_dev.hip", "gda_dsampler.hip", "gda_sampler.cpp", "gda_ppmi.cpp", "gda_smooth.cpp", "gda_comm.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-Wno-unused-value"]


# PYGDA_AMD_MEASUREMENT_AIDS=1: also compile the what-if hooks (PYGDA_AMD_DBG_SKIP, PYGDA_AMD_GEMM_DBG: launches left out /
# kernel phases disabled to read a clock, results WRONG).  The release build -- the default, what build() makes and what
# every test and bench line runs -- contains none of them.
if os.environ.get("PYGDA_AMD_MEASUREMENT_AIDS", "0") == "1":
    FLAGS.append("-DGDA_MEASUREMENT_AIDS")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "gda_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link ``pygda_amd/libgda_hip.so``."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o").replace(".cpp", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
