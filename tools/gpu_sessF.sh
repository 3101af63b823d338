cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh abF "prev new prev new prev new" "C3 C4s C5s"
