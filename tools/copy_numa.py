"""Developer tool: do the device <-> pinned-host copies depend on where the process runs?  Runs tools/copy_mb.py under
three CPU masks: the CPUs local to the GPU's NUMA node, the other CPUs, no mask."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))

if len(sys.argv) > 1 and sys.argv[1] == "probe":
    import torch
    p = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    base = "/sys/bus/pci/devices/" + bdf
    print(bdf, open(base + "/numa_node").read().strip(), open(base + "/local_cpulist").read().strip())
    sys.exit(0)

sys.path.insert(0, os.path.dirname(here))
from opendcvc_amd.dist import _parse_cpulist
bdf, node, cpulist = subprocess.check_output([sys.executable, __file__, "probe"], text=True).split()
allowed = os.sched_getaffinity(0)
local = _parse_cpulist(cpulist) & allowed
remote = allowed - local
print(f"GPU {bdf}: NUMA node {node}, {len(local)} local CPUs of {len(allowed)} allowed; HIP/ROCR_VISIBLE_DEVICES = "
      f"{os.environ.get('HIP_VISIBLE_DEVICES')}/{os.environ.get('ROCR_VISIBLE_DEVICES')}", flush=True)
for name, mask in (("local", local), ("remote", remote), ("no mask", allowed)):
    if not mask:
        continue
    for rep in range(2):
        out = subprocess.run([sys.executable, os.path.join(here, "copy_mb.py")], text=True, capture_output=True,
                             preexec_fn=lambda m=mask: os.sched_setaffinity(0, m)).stdout
        for l in out.splitlines():
            if "index half" in l or "symbols half" in l:
                print(f"[{name:7s} run {rep}] {l}", flush=True)
