cd $GRAFT_REPO_ROOT
NC=$(nproc)
pids=""
for i in $(seq 1 $NC); do (timeout 200 python -c "
while True: pass" &) ; done
sleep 2
for i in $(seq 1 20); do timeout 120 python -m pytest tests/test_mujoco_plugin.py -q -x -m gpu 2>&1 | grep -a -E 'passed|failed|Mismatched|Max abs|^qpos|^qvel|^ee_wpos|^head_z' | tr '\n' ' '; echo; done
