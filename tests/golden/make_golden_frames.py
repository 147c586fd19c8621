"""Regenerates tests/golden/frames_*.json - result frames of the REFERENCE accessors (real lotus code from
/root/reference, `faiss` mapped onto the oracle shim tests/fake_faiss.py) for seeded scenarios.  They travel to the GPU
box, where neither the reference nor faiss exist, and pin lotus_amd.ops + HipVS on the real HIP path
(tests/test_gpu_ops.py).  Run from the repo root: `python tests/golden/make_golden_frames.py`."""
import json
import os
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness  # noqa: E402

lotus = ref_harness.import_lotus()
from lotus.models.rm import RM  # noqa: E402
from lotus.vector_store.faiss_vs import FaissVS  # noqa: E402

import fake_rm  # noqa: E402


def scenario_inputs():
    words = sum(fake_rm.TOPICS.values(), [])
    rng = np.random.default_rng(2026)
    left = [" ".join(rng.choice(words, 3)) for _ in range(80)]
    right = [" ".join(rng.choice(words, 2)) + f" {i}" for i in range(400)]
    dd = [" ".join(rng.choice(words, 4)) for _ in range(120)]
    # near-duplicates only: with exact value repeats the reference's kept-row COUNT depends on which member of a
    # group its hash-ordered set iteration happens to keep (sem_dedup.py:51-88), so it could not be a golden value
    dd += [t + " extra" for t in dd[:25]]
    return dict(left=left, right=right, dedup=dd)


def dump(df):
    return json.loads(df.reset_index().to_json(orient="split"))


def main():
    inp = scenario_inputs()
    out = {"inputs": inp}
    with tempfile.TemporaryDirectory() as td:
        lotus.settings.configure(rm=fake_rm.make_rm(RM), vs=FaissVS())
        df1 = pd.DataFrame({"L": inp["left"]})
        df2 = pd.DataFrame({"R": inp["right"], "keep": np.arange(400) % 5 != 2}).sem_index("R", td + "/r")
        out["join_full_k3"] = dump(df1.sem_sim_join(df2, left_on="L", right_on="R", K=3))
        out["join_filtered_k4"] = dump(df1.sem_sim_join(df2[df2["keep"]], left_on="L", right_on="R", K=4,
                                                        keep_index=True, score_suffix="_s"))
        out["search_k5"] = dump(df2.sem_search("R", "optimization geometry cooking", K=5, return_scores=True))
        out["search_filtered_k3"] = dump(df2[df2["keep"]].sem_search("R", "harry potter history", K=3,
                                                                      return_scores=True))
        dd = pd.DataFrame({"Text": inp["dedup"]}).sem_index("Text", td + "/d")
        kept = dd.sem_dedup("Text", threshold=0.9)
        out["dedup_kept_count"] = int(len(kept))
        cl = pd.DataFrame({"Text": inp["dedup"]}).sem_index("Text", td + "/c").sem_cluster_by("Text", 4, niter=8)
        out["cluster_ids"] = cl["cluster_id"].tolist()
    with open(os.path.join(HERE, "frames_scenarios.json"), "w") as f:
        json.dump(out, f)
    print({k: (len(v["data"]) if isinstance(v, dict) and "data" in v else v if isinstance(v, int) else "...")
           for k, v in out.items() if k != "inputs"})


if __name__ == "__main__":
    main()
