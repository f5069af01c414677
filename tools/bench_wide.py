"""Stage D at the reference's DEFAULT list width (n_neighbors = cells / 5, sampled_fraction = 0.3 -> nrndm = 3000) exactly as bench.py's
extra line runs it: the headline pipeline's pooled matrices and gammas, velocity chain folded in, one launch in column tiles.  Prints the
line; under `rocprofv3 --pmc` (tools/pmc_wide.sh) the LONGEST dispatch of k_cdc_partial_grouped<double ...> is this launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = [sys.argv[0], "--no-extra", "--no-cpu-baseline"] + sys.argv[1:]
a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ["VCY_NO_PROBE"] = "1"
pipe = bench.Pipeline(a, dev, 0, 1, dtype=torch.float64 if a.dtype == "f64" else torch.float32, counts=a.counts)
pipe.step(timed=True)
for _ in range(int(os.environ.get("REPS", 1))):
    line = bench.wide_list_line(a, dev, pipe)
print(json.dumps(line))
