import time, torch, sys
sys.path.insert(0,'.')
import bench
from timeviper_amd.model.llm.nano import NemotronHConfig
t=time.time(); r=bench.cpu_baseline(NemotronHConfig.nemotron_nano_9b_v2()); print(round(time.time()-t,1), r)
