"""Edge-case sequence goldens (SURVEY 8c-6 extended): the REFERENCE models (imported read-only in the build
container, like make_golden.py) on 64x64 I + 2 P frames at both ends of the qp table, without a force-zero
threshold and with a threshold that skips every y symbol.  Output: tests/golden/sequences_edge.json (data only:
per-frame stream hashes / lengths / PSNR).

    python tests/golden/make_golden_edge.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import ref_harness  # noqa: E402


def main():
    DMC, DMCI, L, O, R, S = ref_harness.load()
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    out = {}
    for name, qp, thres, two in (("edge_q0", 0, 0.12, False), ("edge_q63", 63, 0.12, False), ("edge_nothres", 32, None, True),
                                 ("edge_allskip", 32, 100.0, True)):
        mg.THRES = thres
        i_net, p_net = mg.load_models(DMC, DMCI)
        out[name] = mg.run_sequence(i_net, p_net, 64, 64, 3, qp, two, 0, False)[0]
    json.dump(out, open(os.path.join(HERE, "sequences_edge.json"), "w"), indent=1)
    print({k: [f["bytes"] for f in v["frames"]] for k, v in out.items()})


if __name__ == "__main__":
    main()
