import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    z = np.load(os.path.join(HERE, "golden", "encoder_golden.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    cases = []
    for i in range(n):
        name, wseed, B, S, iseed = z[f"case{i}_meta"].tolist()
        cases.append({"name": name, "wseed": int(wseed), "B": int(B), "S": int(S), "iseed": int(iseed),
                      "ids": z[f"case{i}_ids"], "mask": z[f"case{i}_mask"], "emb": z[f"case{i}_emb"],
                      "hidden_valid": z[f"case{i}_hidden_valid"] if f"case{i}_hidden_valid" in z.files else None,
                      "cls_emb": z[f"case{i}_cls_emb"] if f"case{i}_cls_emb" in z.files else None})
    return cases
