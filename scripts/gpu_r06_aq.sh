cd /root/repo
bash scripts/box_probe.sh
OPS=fwd,inv CFGS="c2 c3 c4" bash scripts/gpu_ab.sh "" _noilp _maxilp
