"""micro-probe: random row gather bandwidth vs table size / row size (TLB reach, sector efficiency)"""
import torch, time
dev = torch.device("cuda")
for table_gb in (1, 12, 36):
    for row_words in (16, 192):
        rows = int(table_gb * 1e9 / (row_words * 4))
        x = torch.empty((rows, row_words), dtype=torch.int32, device=dev)
        n = 4_000_000 if row_words == 16 else 1_000_000
        idx = torch.randint(0, rows, (n,), device=dev)
        for _ in range(2):
            y = x[idx]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            y = x[idx]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"table {table_gb} GB rows of {row_words*4} B: gather {n} rows in {dt*1e3:.2f} ms = {n*row_words*4/dt/1e9:.0f} GB/s read")
        del x, y, idx
