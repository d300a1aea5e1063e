import csv,glob,collections
B=1<<20
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/prof_train_pmc_r02/{c}/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for k in ("k_train_chains","k_train_outer","k_train_reduce"):
            if k in n: acc[k].append(float(r["Counter_Value"]))
    for k,v in acc.items(): res[(k,c)]=sum(v)/len(v)
for k in ("k_train_chains","k_train_outer","k_train_reduce"):
    rd=2*res.get((k,"FETCH_SIZE"),0)*1024; wr=res.get((k,"WRITE_SIZE"),0)*1024
    print(k, "read GB %.3f write GB %.3f per launch; per sample %.0f B read %.0f B written" % (rd/1e9, wr/1e9, rd/B, wr/B))
