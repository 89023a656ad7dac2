"""The reference's plugin API: base/base_trainer.py:7-53 (BaseTrainer) and base/base_model.py:6-30
(BaseModel), same constructor arguments, attributes and abstract methods."""
import logging
import time

import numpy as np
import torch
import torch.nn as nn

from .metrics import setup_logger


class BaseTrainer:
    def __init__(self, name, cfg, train_loader, local_rank, logger=None):
        self.cfg = cfg
        self.logger = setup_logger(name + "_train", cfg.OUTPUT_DIR, local_rank) if logger is None else logger
        self.train_loader = train_loader
        self.local_rank = local_rank
        self.start_epoch = 1
        self.distributed = False
        self.lr_data = list()
        self.loss_data = list()
        if torch.cuda.is_available():
            self.with_cuda = True
            device = "cuda"
            if torch.cuda.device_count() > 1:
                self.distributed = True
            torch.cuda.empty_cache()
        else:
            self.logger.warning("Warning: There's no CUDA support on this machine, training is performed on CPU.")
            self.with_cuda = False
            device = "cpu"
        self.device = torch.device(device)
        self.init_params()
        if cfg.resume:
            self.logger.info("Loading checkpoint from {}".format(self.cfg.resume))
            self._load_checkpoint()

    def init_params(self):
        raise NotImplementedError

    def _train_epoch(self, epoch):
        raise NotImplementedError

    def _val_epoch(self, epoch):
        raise NotImplementedError

    def _save_checkpoint(self, epoch, save_path):
        raise NotImplementedError

    def _load_checkpoint(self):
        raise NotImplementedError

    def train(self):
        """Generic epoch loop (base_trainer.py:70-96); ASPPTrainer overrides it like the reference does."""
        best = None
        for epoch in range(self.start_epoch, self.epochs + 1):
            tic = time.time()
            train_log = self._train_epoch(epoch)
            self.logger.info("Epoch {} done in {:.1f}s: {}".format(epoch, time.time() - tic, train_log))
            if epoch % self.val_interval == 0 or epoch == self.epochs:
                val_log = self._val_epoch(epoch)
                score = val_log.get("val_f1")
                if best is None or (score is not None and score > best):
                    self.log = {**train_log, **val_log}
                    self._save_checkpoint(epoch, None)
                    best = score


class BaseModel(nn.Module):
    def __init__(self, config):
        super(BaseModel, self).__init__()
        self.config = config
        self.logger = logging.getLogger(self.__class__.__name__)

    def forward(self, *input):
        raise NotImplementedError

    def summary(self):
        params = sum(np.prod(p.size()) for p in self.parameters() if p.requires_grad)
        self.logger.info("Trainable parameters: {}".format(params))
        self.logger.info(self)
