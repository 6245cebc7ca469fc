/* wcpagc_oracle.c -- TEST INFRASTRUCTURE ONLY.  See wcpagc_oracle.h. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "wcpagc_oracle.h"

#define RB_SIZE ((int)(384000.0 * 8 * 0.01 + 1))       /* wcpAGC.h:30-33 */

void wo_agc_load(wo_agc *a)
{
    double tmp;
    a->attack_buffsize = (int)ceil(a->sample_rate * a->n_tau * a->tau_attack);
    a->in_index = a->attack_buffsize + a->out_index;
    a->attack_mult = 1.0 - exp(-1.0 / (a->sample_rate * a->tau_attack));
    a->decay_mult = 1.0 - exp(-1.0 / (a->sample_rate * a->tau_decay));
    a->fast_decay_mult = 1.0 - exp(-1.0 / (a->sample_rate * a->tau_fast_decay));
    a->fast_backmult = 1.0 - exp(-1.0 / (a->sample_rate * a->tau_fast_backaverage));
    a->onemfast_backmult = 1.0 - a->fast_backmult;
    a->out_target = a->out_targ * (1.0 - exp(-(double)a->n_tau)) * 0.9999;
    a->min_volts = a->out_target / (a->var_gain * a->max_gain);
    a->inv_out_target = 1.0 / a->out_target;
    tmp = log10(a->out_target / (a->max_input * a->var_gain * a->max_gain));
    if (tmp == 0.0) tmp = 1e-16;
    a->slope_constant = (a->out_target * (1.0 - 1.0 / a->var_gain)) / tmp;
    a->inv_max_input = 1.0 / a->max_input;
    tmp = pow(10.0, (a->hang_thresh - 1.0) / 0.125);
    a->hang_level = (a->max_input * tmp + (a->out_target / (a->var_gain * a->max_gain)) * (1.0 - tmp)) * 0.637;
    a->hang_backmult = 1.0 - exp(-1.0 / (a->sample_rate * a->tau_hang_backmult));
    a->onemhang_backmult = 1.0 - a->hang_backmult;
    a->hang_decay_mult = 1.0 - exp(-1.0 / (a->sample_rate * a->tau_hang_decay));
}

void wo_agc_init(wo_agc *a, int run, int mode, int pmode, int sample_rate, double tau_attack, double tau_decay, int n_tau,
                 double max_gain, double var_gain, double fixed_gain, double max_input, double out_targ,
                 double tau_fast_backaverage, double tau_fast_decay, double pop_ratio, int hang_enable,
                 double tau_hang_backmult, double hangtime, double hang_thresh, double tau_hang_decay)
{
    memset(a, 0, sizeof(*a));
    a->run = run; a->mode = mode; a->pmode = pmode; a->sample_rate = (double)sample_rate;
    a->tau_attack = tau_attack; a->tau_decay = tau_decay; a->n_tau = n_tau; a->max_gain = max_gain; a->var_gain = var_gain;
    a->fixed_gain = fixed_gain; a->max_input = max_input; a->out_targ = out_targ;
    a->tau_fast_backaverage = tau_fast_backaverage; a->tau_fast_decay = tau_fast_decay; a->pop_ratio = pop_ratio;
    a->hang_enable = hang_enable; a->tau_hang_backmult = tau_hang_backmult; a->hangtime = hangtime;
    a->hang_thresh = hang_thresh; a->tau_hang_decay = tau_hang_decay;
    /* calc_wcpagc, wcpAGC.c:29-47 */
    a->ring_buffsize = RB_SIZE;
    a->out_index = -1;
    a->ring = (double *)calloc((size_t)RB_SIZE * 2, sizeof(double));
    a->abs_ring = (double *)calloc((size_t)RB_SIZE, sizeof(double));
    wo_agc_load(a);
}

void wo_agc_free(wo_agc *a) { free(a->ring); free(a->abs_ring); a->ring = a->abs_ring = NULL; }

void wo_agc_set_mode(wo_agc *a, int mode)
{
    switch (mode) {
    case 0: a->mode = 0; wo_agc_load(a); break;
    case 1: a->mode = 1; a->hangtime = 2.000; a->tau_decay = 2.000; wo_agc_load(a); break;
    case 2: a->mode = 2; a->hangtime = 1.000; a->tau_decay = 0.500; wo_agc_load(a); break;
    case 3: a->mode = 3; a->hang_thresh = 1.0; a->hangtime = 0.000; a->tau_decay = 0.250; wo_agc_load(a); break;
    case 4: a->mode = 4; a->hang_thresh = 1.0; a->hangtime = 0.000; a->tau_decay = 0.050; wo_agc_load(a); break;
    default: a->mode = 5; break;
    }
}

void wo_agc_exec(wo_agc *a, double *buf, int size)
{
    int i, j, k;
    double mult, o0, o1, abs_out;
    if (!a->run) return;
    if (a->mode == 0) {
        for (i = 0; i < size; i++) { buf[2 * i] = a->fixed_gain * buf[2 * i]; buf[2 * i + 1] = a->fixed_gain * buf[2 * i + 1]; }
        return;
    }
    for (i = 0; i < size; i++) {
        if (++a->out_index >= a->ring_buffsize) a->out_index -= a->ring_buffsize;
        if (++a->in_index >= a->ring_buffsize) a->in_index -= a->ring_buffsize;
        o0 = a->ring[2 * a->out_index]; o1 = a->ring[2 * a->out_index + 1];
        abs_out = a->abs_ring[a->out_index];
        a->ring[2 * a->in_index] = buf[2 * i]; a->ring[2 * a->in_index + 1] = buf[2 * i + 1];
        if (a->pmode == 0)
            a->abs_ring[a->in_index] = fmax(fabs(a->ring[2 * a->in_index]), fabs(a->ring[2 * a->in_index + 1]));
        else
            a->abs_ring[a->in_index] = sqrt(a->ring[2 * a->in_index] * a->ring[2 * a->in_index] +
                                            a->ring[2 * a->in_index + 1] * a->ring[2 * a->in_index + 1]);
        a->fast_backaverage = a->fast_backmult * abs_out + a->onemfast_backmult * a->fast_backaverage;
        a->hang_backaverage = a->hang_backmult * abs_out + a->onemhang_backmult * a->hang_backaverage;
        if ((abs_out >= a->ring_max) && (abs_out > 0.0)) {
            a->ring_max = 0.0;
            k = a->out_index;
            for (j = 0; j < a->attack_buffsize; j++) {
                if (++k == a->ring_buffsize) k = 0;
                if (a->abs_ring[k] > a->ring_max) a->ring_max = a->abs_ring[k];
            }
        }
        if (a->abs_ring[a->in_index] > a->ring_max) a->ring_max = a->abs_ring[a->in_index];
        if (a->hang_counter > 0) --a->hang_counter;
        switch (a->state) {
        case 0:
            if (a->ring_max >= a->volts) a->volts += (a->ring_max - a->volts) * a->attack_mult;
            else if (a->volts > a->pop_ratio * a->fast_backaverage) { a->state = 1; a->volts += (a->ring_max - a->volts) * a->fast_decay_mult; }
            else if (a->hang_enable && (a->hang_backaverage > a->hang_level)) {
                a->state = 2; a->hang_counter = (int)(a->hangtime * a->sample_rate); a->decay_type = 1;
            } else { a->state = 3; a->volts += (a->ring_max - a->volts) * a->decay_mult; a->decay_type = 0; }
            break;
        case 1:
            if (a->ring_max >= a->volts) { a->state = 0; a->volts += (a->ring_max - a->volts) * a->attack_mult; }
            else if (a->volts > a->save_volts) a->volts += (a->ring_max - a->volts) * a->fast_decay_mult;
            else if (a->hang_counter > 0) a->state = 2;
            else if (a->decay_type == 0) { a->state = 3; a->volts += (a->ring_max - a->volts) * a->decay_mult; }
            else { a->state = 4; a->volts += (a->ring_max - a->volts) * a->hang_decay_mult; }
            break;
        case 2:
            if (a->ring_max >= a->volts) { a->state = 0; a->save_volts = a->volts; a->volts += (a->ring_max - a->volts) * a->attack_mult; }
            else if (a->hang_counter == 0) { a->state = 4; a->volts += (a->ring_max - a->volts) * a->hang_decay_mult; }
            break;
        case 3:
            if (a->ring_max >= a->volts) { a->state = 0; a->save_volts = a->volts; a->volts += (a->ring_max - a->volts) * a->attack_mult; }
            else a->volts += (a->ring_max - a->volts) * a->decay_mult;
            break;
        case 4:
            if (a->ring_max >= a->volts) { a->state = 0; a->save_volts = a->volts; a->volts += (a->ring_max - a->volts) * a->attack_mult; }
            else a->volts += (a->ring_max - a->volts) * a->hang_decay_mult;
            break;
        }
        if (a->volts < a->min_volts) a->volts = a->min_volts;
        a->gain = a->volts * a->inv_out_target;
        mult = (a->out_target - a->slope_constant * fmin(0.0, log10(a->inv_max_input * a->volts))) / a->volts;
        buf[2 * i] = o0 * mult; buf[2 * i + 1] = o1 * mult;
    }
}
