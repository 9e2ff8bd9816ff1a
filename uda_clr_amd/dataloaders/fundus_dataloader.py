"""``FundusSegmentation`` - drop-in for dataloaders/fundus_dataloader.py:11-83.

Layout (Appendix C of SURVEY.md): <base_dir>/<dataset>/<split>/ROIs/image/*.png with the mask at the same
path where every 'image' is replaced by 'mask'.  All images are decoded into memory at construction; a
sample is the dict {'image', 'label', 'img_name'} handed to ``transform``."""
import os
from glob import glob

from PIL import Image
from torch.utils.data import Dataset

from .mypath import Path


class FundusSegmentation(Dataset):
    def __init__(self, base_dir=Path.db_root_dir('fundus'), dataset='refuge', split='train', testid=None, transform=None):
        self._base_dir = base_dir
        self.split = split
        self.transform = transform
        self._image_dir = os.path.join(base_dir, dataset, split, 'ROIs', 'image')
        self.image_list = [{'image': p, 'label': p.replace('image', 'mask'), 'id': testid}
                           for p in glob(self._image_dir + '/*.png')]
        self.image_pool, self.label_pool, self.img_name_pool = [], [], []
        for item in self.image_list:
            self.image_pool.append(Image.open(item['image']).convert('RGB'))
            target = Image.open(item['label'])
            self.label_pool.append(target.convert('L') if target.mode == 'RGB' else target)
            self.img_name_pool.append(item['image'].split('/')[-1])
        print('Number of images in {}: {:d}'.format(split, len(self.image_list)))

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, index):
        sample = {'image': self.image_pool[index], 'label': self.label_pool[index], 'img_name': self.img_name_pool[index]}
        return self.transform(sample) if self.transform is not None else sample

    def __str__(self):
        return 'Fundus(split=' + str(self.split) + ')'
