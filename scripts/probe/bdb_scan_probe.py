import dbm.ndbm as ndbm, os, time, numpy as np, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bigsi_amd import bdb, _lib
d='/dev/shm/bdbt'; os.makedirs(d,exist_ok=True)
rng=np.random.default_rng(0)
rb=62500; m=100000
base=[rng.integers(0,256,size=rb,dtype=np.uint8).tobytes() for _ in range(16)]
db=ndbm.open(d+'/s','n')
for i in range(m): db["%d:bitarray"%i]=base[i%16]
db["number_of_rows:int"]=str(m)
db.close()
_lib.lib()
sz=os.path.getsize(d+'/s.db')
print("file GB", sz/1e9)
L=_lib.lib()
for th in (16,48,8,16,32,48,64,16):
    need, rows, widest = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    buf=np.zeros(1<<20,np.uint8)
    t=time.time(); rc=L.bigsi_hip_bdb_small_records((d+'/s.db').encode(), _lib.ptr(buf), buf.size, C.byref(need), C.byref(rows), C.byref(widest), th); dt=time.time()-t
    print("threads",th, round(dt,3), "s", round(sz/dt/1e9,1),"GB/s", rows.value)
import shutil; shutil.rmtree(d)
