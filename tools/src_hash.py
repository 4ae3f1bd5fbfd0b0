"""Hash of the kernel sources a PMC traffic figure belongs to (profiles/pmc_traffic.json entries carry it; bench.py reports
`traffic: null` when the sources have changed since the counters were collected)."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = "unidom_amd/csrc"
GROUPS = {   # kernel-name prefix (or "large_path:" key) -> the sources its code is compiled from
    "cloth": ["cloth.hip", "cloth_ref.hip", "cloth_ref_order.h", "cloth_v2.hip", "cloth_fast.hip", "cloth_cluster_fwd.hip", "cloth_cluster_bwd.hip", "cloth_cluster.h", "cloth_v2_force.h",
              "cloth_fast_adj.h", "cloth_common.h", "exact_math.h", "common.h"],
    "chamfer": ["env_glue.hip", "common.h"],
    "pnp": ["env_glue.hip", "common.h"],
    "mpm_step": ["mpm.hip", "mpm_device.h", "common.h"],
    "mpm_focus": ["env_glue.hip", "common.h"],
    "mpm_finish": ["env_glue.hip", "common.h"],
    "large_path": ["mpm_large.hip", "mpm_cluster.h", "mpm_large.h", "mpm_device.h", "mpm_collide.h", "common.h"],
    "lg_": ["mpm_large.hip", "mpm_cluster.h", "mpm_large.h", "mpm_device.h", "mpm_collide.h", "common.h"],
    "clm_": ["mpm_large.hip", "mpm_cluster.h", "mpm_large.h", "mpm_device.h", "mpm_collide.h", "common.h"],
    "plb": ["plb.hip", "plb_adj.hip", "plb_cluster.hip", "plb_common.h", "plb_device.h", "common.h"],
    "pcl_": ["plb.hip", "plb_adj.hip", "plb_cluster.hip", "plb_common.h", "plb_device.h", "common.h"],
}


def group_of(kernel):
    for k in GROUPS:
        if kernel.startswith(k):
            return k
    return None


def sha16(kernel, rev=None):
    """sha256[:16] over the group's files (name + bytes); files that do not exist are skipped.  rev: a git revision to
    hash instead of the working tree (used once to stamp figures measured on an earlier commit)."""
    g = group_of(kernel)
    if g is None:
        return None
    h = hashlib.sha256()
    for f in GROUPS[g]:
        path = f"{CSRC}/{f}"
        if rev:
            r = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{path}"], capture_output=True)
            if r.returncode != 0:
                continue
            data = r.stdout
        else:
            if not os.path.exists(os.path.join(ROOT, path)):
                continue
            data = open(os.path.join(ROOT, path), "rb").read()
        h.update(f.encode() + b"\0" + data)
    return h.hexdigest()[:16]
