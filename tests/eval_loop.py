"""TEST HARNESS (not part of the product): a driver with the contract of the reference's detection/tools/eval_utils.py:28-146
(recall bookkeeping :13-25, per-batch ``load_data_to_gpu`` -> ``model(batch_dict)`` -> ``generate_prediction_dicts``,
``result.pkl``, ``dataset.evaluation``) for machines where /root/reference is absent (the GPU box).  Where the tree exists the
reference's own file is loaded by path and runs unchanged against the shim packages; tests/test_shim.py drives both over the
same loader and compares the pickles."""
import pickle
import time

import torch

from detzero_utils import common_utils
from detzero_det.models import load_data_to_gpu


def statistics_info(cfg, ret_dict, metric, disp_dict):
    thr = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST
    for t in thr:
        metric['recall_roi_%s' % str(t)] += ret_dict.get('roi_%s' % str(t), 0)
        metric['recall_rcnn_%s' % str(t)] += ret_dict.get('rcnn_%s' % str(t), 0)
    metric['gt_num'] += ret_dict.get('gt', 0)
    for t in (thr[0], thr[-1]):
        disp_dict['recall_%s' % str(t)] = '(%d, %d) / %d' % (metric['recall_roi_%s' % str(t)], metric['recall_rcnn_%s' % str(t)], metric['gt_num'])


def eval_one_epoch(cfg, model, dataloader, epoch_id, logger, dist_test=False, save_to_file=False, result_dir=None, save_tb=True):
    result_dir.mkdir(parents=True, exist_ok=True)
    final_output_dir = result_dir / 'data'
    if save_to_file:
        final_output_dir.mkdir(parents=True, exist_ok=True)
    thr = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST
    metric = {'gt_num': 0}
    for t in thr:
        metric['recall_roi_%s' % str(t)] = 0
        metric['recall_rcnn_%s' % str(t)] = 0
    dataset = dataloader.dataset
    class_names = dataset.class_names
    det_annos = []
    logger.info('*************** EPOCH %s EVALUATION *****************' % epoch_id)
    model.eval()          # weights are replicated and nothing is reduced in eval: no DistributedDataParallel wrapper is needed
    start = time.time()
    for i, batch_dict in enumerate(dataloader):
        batch_dict['eval_iter'] = i
        load_data_to_gpu(batch_dict)
        with torch.no_grad():
            pred_dicts, ret_dict = model(batch_dict)[:2]
        statistics_info(cfg, ret_dict, metric, {})
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, class_names,
                                                       output_path=final_output_dir if save_to_file else None)
    if dist_test:
        rank, world_size = common_utils.get_dist_info()
        det_annos = common_utils.merge_results_dist(det_annos, len(dataset), tmpdir=result_dir / 'tmpdir')
        metric = common_utils.merge_results_dist([metric], world_size, tmpdir=result_dir / 'tmpdir')
    logger.info('Generate label finished(sec_per_example: %.4f second).' % ((time.time() - start) / max(len(dataloader.dataset), 1)))
    if cfg.LOCAL_RANK != 0:
        return {}
    if dist_test:
        for key in metric[0]:
            for k in range(1, world_size):
                metric[0][key] += metric[k][key]
        metric = metric[0]
    ret = {}
    for t in thr:
        ret['recall/roi_%s' % str(t)] = metric['recall_roi_%s' % str(t)] / max(metric['gt_num'], 1)
        ret['recall/rcnn_%s' % str(t)] = metric['recall_rcnn_%s' % str(t)] / max(metric['gt_num'], 1)
        logger.info('recall_rcnn_%s: %f' % (t, ret['recall/rcnn_%s' % str(t)]))
    n_obj = sum(len(a['name']) for a in det_annos)
    logger.info('Average predicted number of objects(%d samples): %.3f' % (len(det_annos), n_obj / max(1, len(det_annos))))
    with open(result_dir / 'result.pkl', 'wb') as f:
        pickle.dump(det_annos, f)
    result_str, result_dict = dataset.evaluation(det_annos, class_names, eval_metric=cfg.MODEL.POST_PROCESSING.EVAL_METRIC,
                                                 output_path=final_output_dir)
    logger.info(result_str)
    ret.update(result_dict)
    logger.info('Result is save to %s' % result_dir)
    return ret
