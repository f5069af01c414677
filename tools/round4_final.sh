#!/bin/bash
# the round's closing call: whole GPU suite + smoke(), then the profile passes, the driver-flag bench line and the side measurements
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{ time timeout 1500 python -m pytest -q -m gpu tests/ ; } > gpurun_out/r04_tests_all.log 2>&1
tail -6 gpurun_out/r04_tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/round4_measure_all.sh
