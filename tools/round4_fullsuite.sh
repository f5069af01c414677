#!/bin/bash
# the driver's round-end GPU tier: the whole GPU suite, smoke(), one default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{ time timeout 3000 python -m pytest -q -m gpu tests/ ; } > gpurun_out/r04_tests_all.log 2>&1
tail -8 gpurun_out/r04_tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
{ time python bench.py ; } 2>&1 | tail -5 | cut -c1-300
