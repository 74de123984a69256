#!/bin/bash
# VERDICT r2 next #1c: can the GPU lease reach a package index (to build real Ceres for tests/test_ref_ceres.py)?
# Every attempt is bounded; output goes to gpurun_out/net_probe.log.
out=gpurun_out/net_probe.log
mkdir -p gpurun_out
{
echo "== date: $(date -u)"; nproc; 
echo "== on disk"; find / -xdev \( -iname "*ceres*" -o -iname "Eigen" -o -iname "glog" -o -iname "libglog*" -o -iname "*suitesparse*" -o -iname "libcholmod*" \) 2>/dev/null | grep -v -E "^/proc|gpurun|/root/repo|graft" | head -20
echo "== pip download pyceres"; timeout 25 pip download --no-deps -d /tmp/w pyceres 2>&1 | tail -3
echo "== pip download ceres-solver / eigen"; timeout 25 pip download --no-deps -d /tmp/w ceres-solver 2>&1 | tail -2
echo "== pip config"; pip config list 2>&1 | head
echo "== curl pypi"; timeout 15 curl -sS -m 10 -o /dev/null -w "%{http_code}\n" https://pypi.org/simple/pyceres/ 2>&1 | tail -1
echo "== curl github"; timeout 15 curl -sS -m 10 -o /dev/null -w "%{http_code}\n" https://github.com/ceres-solver/ceres-solver/archive/refs/tags/2.0.0.tar.gz 2>&1 | tail -1
echo "== apt-get"; timeout 30 apt-get update 2>&1 | tail -3
echo "== apt-cache policy libceres-dev"; apt-cache policy libceres-dev libeigen3-dev libgoogle-glog-dev 2>&1 | head -12
echo "== conda"; (which conda mamba micromamba; timeout 30 conda search -c conda-forge ceres-solver 2>&1 | tail -3)
echo "== dns"; getent hosts pypi.org github.com conda.anaconda.org 2>&1 | head
echo "== env proxies"; env | grep -i -E "proxy|index" | head
} > $out 2>&1
echo done
