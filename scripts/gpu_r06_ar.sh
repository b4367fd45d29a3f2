cd /root/repo
bash scripts/box_probe.sh
PREC=bx3 OPS=fwd,inv CFGS="c2" bash scripts/gpu_ab.sh "" _st2 _st3
