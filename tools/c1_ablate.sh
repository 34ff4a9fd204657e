cd $GRAFT_REPO_ROOT
for d in "" "-DC1_ABL=1" "-DC1_ABL=2" "-DC1_ABL=3"; do
  touch glare_amd/csrc/conv1x1.hip; GLARE_DEFS="$d" python glare_amd/csrc/build.py > /dev/null 2>&1 || echo build failed
  echo "== [$d]"; KB_CONV="512->512 1x1" KB_REPS=10 python tools/kbench.py conv 2>&1 | grep "weight-stationary"
done
touch glare_amd/csrc/conv1x1.hip; python glare_amd/csrc/build.py > /dev/null 2>&1
