import sys,time,os,subprocess
code='''
import sys,time,os
sys.path.insert(0,'oracle'); sys.path.insert(0,'video-super-resolution-library_amd')
import oracle_py as O, synth
y=synth.natural_y(1920,1080)
p1=O.make_pass(O.Model('filters_2x/filters_highres',8,1),8)
O.process_y(y[:128,:256],512,256,p1)
t=time.time(); o=O.process_y(y,3840,2160,p1); t1=time.time()-t
t=time.time(); o=O.process_y(y,3840,2160,p1); t2=time.time()-t
print(os.environ.get('OMP_NUM_THREADS'),"threads:",round(t1,3),round(t2,3),"s", os.environ.get('OMP_PROC_BIND'))
'''
for t in (8,16,32,64,128,256):
    for bind in ("false","close","spread"):
        env=dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND=bind)
        print(subprocess.run([sys.executable,"-c",code],env=env,capture_output=True,text=True).stdout.strip())
