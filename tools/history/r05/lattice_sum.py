import numpy as np
rng=np.random.RandomState(1)
worst=0
for trial in range(4000):
    spt=rng.uniform(3.0,40.0); ang=rng.uniform(0,2*np.pi); cx,cy=rng.uniform(0,1,2)
    R=int(2*spt)+3
    ys,xs=np.mgrid[-R:R+1,-R:R+1]
    dx=xs-cx; dy=ys-cy
    c,s=np.cos(ang),np.sin(ang)
    nx=(c*dx+s*dy)/spt; ny=(c*dy-s*dx)/spt
    w=np.clip(1-np.abs(nx),0,None)*np.clip(1-np.abs(ny),0,None)
    S=w.sum()
    worst=max(worst,S/((spt+1)**2))
    if trial<3: print(spt,S,spt**2)
print("max lattice sum / (spt+1)^2 =",worst)
