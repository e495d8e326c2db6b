import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from orb_slam3_fast_amd import synth
from oracle import oracle_py as O
cols = synth.make_vocabulary_bfs(10, 6, seed=1)
ov = O.Vocabulary(10, 6, *cols)
t = time.time(); ov.save("/tmp/ORBvoc_synth.txt"); print("save %.1fs, %.0f MB" % (time.time() - t, os.path.getsize("/tmp/ORBvoc_synth.txt") / 1e6))
t = time.time(); ov2 = O.Vocabulary(path="/tmp/ORBvoc_synth.txt"); print("oracle load %.1fs" % (time.time() - t), ov2.n_nodes, ov2.n_words)
if len(sys.argv) > 1:
    import orb_slam3_fast_amd as orbx
    t = time.time(); v = orbx.ORBVocabulary(path="/tmp/ORBvoc_synth.txt"); print("device load %.1fs" % (time.time() - t), v.n_nodes, v.n_words)
    f = synth.vocabulary_features(cols, 1500, 3)
    a, b = v.transform(f, 4), ov2.transform(f, 4)
    print("same", all(np.array_equal(x, y) for x, y in zip(a[0] + a[1], b[0] + b[1])))
