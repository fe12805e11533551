# round 4, GPU call 21: where the per-epoch K-means refresh goes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 500 python tools/kmeans_profile.py > gpurun_out/r4_kmeans21.txt 2>&1; head -60 gpurun_out/r4_kmeans21.txt | cut -c1-200
