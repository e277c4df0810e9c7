"""Golden vectors of the EVALUATOR FORMATS from the IMPORTED REFERENCE (this container only).
Run:  python tests/golden/make_golden_eval.py   ->  tests/golden/eval_formats.pt

Drives the reference's own `valid_one_epoch_cl_single_gpu` (MQ/libs/utils/train_utils.py:1016-1173) and `final_validate`
(:1176-1360) over two validation tasks of two clips each with the model of the "xl" golden case (state in model_xl.pt).  The two things
those functions hand to code outside the hot path are recorded:
  * the `results` dict every evaluator call receives ({'video-id': [...], 't-start' / 't-end' / 'label' / 'score': numpy
    arrays}, :1060-1097), through a recording evaluator;
  * the ActivityNet-style JSON they write for the retrieval metric ({"version": "1.0", "external_data": "", "results":
    {video_id: [{"segment": [s, e], "score", "label": <class NAME>}]}}, :1118-1134), through a stand-in for
    `evaluation_retrieval` that reads the file back (the metric itself needs the Ego4D annotation pickle: outside the path);
plus the tuples the functions return for the stand-in metric values and final_validate's per-task bookkeeping lists."""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402


class _Log:
    def info(self, *a, **k):
        pass


class ValTasks:
    def get_valSet_by_taskNum(self, n):
        return [([[c] for c in cases.eval_clips(k)], 3 + k) for k in range(n)]


class Recorder:
    def __init__(self):
        self.calls = []

    def evaluate(self, results, current_task_id=None, verbose=False):
        self.calls.append({k: (list(v) if k == 'video-id' else np.array(v)) for k, v in results.items()})
        m = cases.eval_fake_map(results)
        return np.array([m] * 5), m, np.linspace(0.1, 0.5, 5)


def main():
    libs = ref_import.setup(extra_xlnet=((32, 4), (64, 4), (128, 4), (2304, 16)))
    from libs.modeling import make_meta_arch
    import libs.utils.train_utils as TU
    gold = torch.load(os.path.join(HERE, 'model_xl.pt'), weights_only=False)
    cfg = ref_import.make_cfg(libs, **gold['overrides'])
    torch.manual_seed(1)
    model = make_meta_arch(cfg['model_name'], **cfg['model'])
    model.load_state_dict(gold['state_dict'])
    jsons = []

    def fake_retrieval(gt, pred, subset, tiou, use_cl=False, current_task_id=None):
        with open(pred) as f:
            obj = json.load(f)
        jsons.append(obj)
        return cases.eval_fake_recall(obj)
    TU.evaluation_retrieval = fake_retrieval
    cwd = os.getcwd()
    os.chdir('/tmp')                                           # the function writes retrieval_json/ under the CWD
    try:
        rec = Recorder()
        ret = TU.valid_one_epoch_cl_single_gpu(ValTasks(), model, 0, 1, evaluator=rec, output_file='g', logger=_Log(),
                                               dataset_name='ego4d_cl')
        torch.set_grad_enabled(True)
        out = {'valid': {'results': rec.calls, 'json': copy.deepcopy(jsons), 'ret': [float(x) for x in ret]}}
        jsons.clear()
        rec = Recorder()
        rl, ml = {'val': [0.9]}, {'val': [0.8]}                # what task 0's final validation left behind
        ret = TU.final_validate(ValTasks(), model, 0, 1, evaluator=rec, output_file='g', logger=_Log(), dataset_name='ego4d_cl',
                                list_val_recall_ii=rl, list_val_mAP_ii=ml, type_val='val')
        torch.set_grad_enabled(True)
        out['final'] = {'results': rec.calls, 'json': copy.deepcopy(jsons), 'ret': None if ret is None else [float(x) for x in ret],
                        'recall_list': [float(x) for x in rl['val']], 'map_list': [float(x) for x in ml['val']]}
    finally:
        os.chdir(cwd)
    names = {}
    for j in out['valid']['json']:
        for vid, rows in j['results'].items():
            pass
    # class-name table rows the JSONs used (label id -> name), recovered by pairing rows with the results dict
    for call, j in zip(out['valid']['results'], out['valid']['json']):
        pos = {}
        for vid, lab in zip(call['video-id'], call['label']):
            k = pos.get(vid, 0)
            names[int(lab)] = j['results'][vid][k]['label']
            pos[vid] = k + 1
    out['idx_classes'] = names
    path = os.path.join(HERE, 'eval_formats.pt')
    torch.save(out, path)
    print('valid ret', out['valid']['ret'], 'final ret', out['final']['ret'], out['final']['recall_list'], out['final']['map_list'],
          [len(c['video-id']) for c in out['valid']['results']], len(names), '%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
