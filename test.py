"""Evaluation entry point: same flags as the reference's test.py:44-72
(`-cfg FILE [--saveres] [-c renders/cityscapes.json] KEY VAL ...`)."""
import argparse

import torch

from core.configs import cfg
from core.datasets.build import build_collate_fn, build_dataset
from core.testers.aspp_tester import ASPPTester
from core.utils.utility import load_json, setup_logger

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def test(cfg, config, args):
    name = config["name"]
    logger = setup_logger(name + "_test", cfg.OUTPUT_DIR, None)
    logger.info("#" * 20 + " Start Testing " + "#" * 20)
    logger.info("INPUT_SIZE_TEST: {}".format(cfg.INPUT.INPUT_SIZE_TEST))
    data = build_dataset(cfg, mode="test", is_source=False)
    loader = torch.utils.data.DataLoader(data, batch_size=cfg.TEST.BATCH_SIZE, shuffle=False, num_workers=2, pin_memory=True,
                                         collate_fn=build_collate_fn(cfg), sampler=None)
    if name.startswith("pranet"):                            # reference test.py:35-36
        from core.testers.pranet_tester import PranetTester
        tester = PranetTester(cfg, device, loader, logger)
    elif name.startswith("gald"):                            # reference test.py:39-40
        from core.testers.gald_tester import GALDTester
        tester = GALDTester(cfg, device, loader, logger, config["palette"], saveres=args.saveres, trainid2name=config.get("trainid2name"))
    elif not name.startswith("aspp"):
        raise NotImplementedError("tester %r: 'aspp*', 'pranet*' and 'gald*' are on the MI355X engine" % name)
    else:
        tester = ASPPTester(cfg, device, loader, logger, config["palette"], config["trainid2name"], saveres=args.saveres)
    if cfg.resume:
        tester._load_checkpoint()
    else:
        logger.warning("cfg.resume is empty: evaluating freshly initialised weights")
    return tester.test()


def main():
    parser = argparse.ArgumentParser(description="MI355X Semantic Segmentation Testing")
    parser.add_argument("-cfg", "--config-file", default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument("--saveres", action="store_true", help="save the result")
    parser.add_argument("-c", "--config_path", default="renders/cityscapes.json", help="path to config")
    parser.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    config = load_json(args.config_path)
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    print("Loaded configuration file {}".format(args.config_file))
    test(cfg, config, args)


if __name__ == "__main__":
    main()
