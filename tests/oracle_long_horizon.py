"""Long-horizon check of the learning behaviour on the CPU oracle (TEST INFRASTRUCTURE; see profiles/learning_r02a_oracle_check.json).
    python tests/oracle_long_horizon.py SEED EPISODES_PER_PHASE {Malicious|Faulty|Greedy|Cooperative} H
"""
import sys, time, json
import os; _T = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(_T)); sys.path.insert(0, _T)
import numpy as np
from oracle import rpbcac_oracle as O, mlp_np as M
seed=int(sys.argv[1]); n_ep=int(sys.argv[2]); scen=sys.argv[3]; H=int(sys.argv[4])
labels=["Cooperative"]*4+[scen]
IN=[[0,1,2,3],[1,2,3,4],[2,3,4,0],[3,4,0,1],[4,0,1,2]]
args={"n_agents":5,"agent_label":labels,"in_nodes":IN,"n_actions":5,"n_states":2,"n_episodes":n_ep,"max_ep_len":20,"n_ep_fixed":50,"n_epochs":10,"slow_lr":0.002,"fast_lr":0.01,"batch_size":200,"buffer_size":2000,"gamma":0.9,"H":H,"common_reward":False,"random_seed":seed}
rng=np.random.default_rng([2, seed])
def glorot(i,h,o): return M.init_mlp(rng,i,h,o)
agents=[O.make_agent(l, glorot(10,20,5), glorot(10,20,1), glorot(15,20,1), 0.002,0.01,0.9,H) for l in labels]
goal=np.random.RandomState(seed).randint(0,5,size=(5,2))
env=O.GridWorldOracle(5,5,5,goal,None,True,True,rng_mode="device",seed=seed)
t=time.time()
w,df1=O.train(env,agents,args,rng_mode="device")
print("phase1", time.time()-t, df1["True_team_returns"].to_numpy()[-500:].mean(), flush=True)
# phase 2: fresh Adam + empty replay (agents keep weights)
for a in agents:
    for attr in ("adam","actor_adam","opt"):
        if hasattr(a,attr):
            st=getattr(a,attr)
            if hasattr(st,"t"): st.t=0; st.m=[np.zeros_like(x) for x in st.m]; st.v=[np.zeros_like(x) for x in st.v]
w,df2=O.train(env,agents,args,rng_mode="device")
print("phase2", time.time()-t, df2["True_team_returns"].to_numpy()[-500:].mean(), flush=True)
json.dump({"seed":seed,"phase1":float(df1["True_team_returns"].to_numpy()[-500:].mean()),"phase2":float(df2["True_team_returns"].to_numpy()[-500:].mean())}, open("/tmp/oracle_long_%s_H%d_%d.json"%(scen,H,seed),"w"))
