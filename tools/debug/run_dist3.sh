cd /root/repo
python - <<'PY' > /tmp/dist3.py
import re
s = open('tests/test_model_gpu.py').read()
a = s.index('_DIST_SCRIPT = r"""') + len('_DIST_SCRIPT = r"""')
b = s.index('"""', a)
print(s[a:b])
PY
SAM_REPO=/root/repo MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 SAM_TEST_THREE_GROUPS=1 python /tmp/dist3.py 2>&1 | grep -v "^$" | tail -25
